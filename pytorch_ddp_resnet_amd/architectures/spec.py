"""
``architecture_spec`` grammar, as implemented by the reference (not as its docstring says).

Behaviour mirrored from /root/reference/resnet/architectures/resnet.py:16-22 (integer extraction) and :122-158
(prefix dispatch, running channel count, "a stack downsamples iff the previous token starts with the same
letter").  Error behaviour kept: an unknown token raises ``ValueError("Unknown component in architecture spec.")``
(resnet.py:156); a malformed token raises ``AttributeError`` from the failed regex match (resnet.py:18-19).
"""
import re
from dataclasses import dataclass
from typing import List, Tuple


@dataclass(frozen=True)
class Component:
    kind: str                 # conv | maxpool | avgpool | basic | bottleneck | norm | act | fc
    args: Tuple[int, ...] = ()
    cin: int = 0
    cout: int = 0
    down: bool = False
    depth: int = 0


def _ints(token: str, count: int) -> Tuple[int, ...]:
    m = re.match(r"([a-z]+)" + ",".join(r"([0-9]+)" for _ in range(count)), token)
    return tuple(int(g) for g in m.groups()[1:])       # AttributeError on a malformed token, like the reference


# dispatch order matters: 'ap' must not be reached by 'a', nor 'mp' by anything else (resnet.py:126-154)
_PREFIXES = (('c', 'conv'), ('mp', 'maxpool'), ('ap', 'avgpool'), ('r', 'basic'), ('b', 'bottleneck'),
             ('n', 'norm'), ('a', 'act'), ('f', 'fc'))


def parse_spec(spec: str) -> List[Component]:
    tokens = spec.split()
    comps: List[Component] = []
    channels = None
    for pos, tok in enumerate(tokens):
        kind = next((k for p, k in _PREFIXES if tok.startswith(p)), None)
        if kind is None:
            raise ValueError("Unknown component in architecture spec.")
        if kind == 'conv':
            i, o, k, s, p = _ints(tok, 5)
            comps.append(Component('conv', (k, s, p), cin=i, cout=o))
            channels = o
        elif kind in ('maxpool', 'avgpool'):
            comps.append(Component(kind, _ints(tok, 3)))
        elif kind in ('basic', 'bottleneck'):
            down = tokens[pos - 1].startswith(tok[0])      # pos-1 == -1 wraps to the last token, as in the reference
            cout = 2 * channels if down else channels
            comps.append(Component(kind, cin=channels, cout=cout, down=down, depth=_ints(tok, 1)[0]))
            channels = cout
        elif kind == 'norm':
            comps.append(Component('norm', cin=channels, cout=channels))
        elif kind == 'act':
            comps.append(Component('act'))
        else:
            i, o = _ints(tok, 2)
            comps.append(Component('fc', cin=i, cout=o))
    return comps


def block_convs(kind: str, cin: int, down: bool, preact: bool):
    """-> ([(cin, cout, k, stride, pad)], [bn feature counts], cout) for one block
    (residual_block.py:26-61 basic; :120-165 bottleneck: width C/4, or C/2 in a downsampling block, stride on the 3x3)."""
    cout = 2 * cin if down else cin
    s = 2 if down else 1
    if kind == 'basic':
        convs = [(cin, cout, 3, s, 1), (cout, cout, 3, 1, 1)]
        norms = [cin if preact else cout, cout]
    else:
        mid = cin // 2 if down else cin // 4
        convs = [(cin, mid, 1, 1, 0), (mid, mid, 3, s, 1), (mid, cout, 1, 1, 0)]
        norms = [cin if preact else mid, mid, mid if preact else cout]
    return convs, norms, cout
