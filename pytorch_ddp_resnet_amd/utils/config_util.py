"""
YAML run configuration.  Mirrors /root/reference/resnet/utils/config_util.py:6-28 (``ConfigParser(defaults)``,
``read(path, verbose)``, ``get``, ``[]``, ``items``; a missing key raises ``KeyError`` from ``get`` exactly like the
reference) and fixes its defect: the reference subclasses ``dict`` but never fills it, so ``f(**config)``
passes no keyword arguments; here the mapping itself is populated, so ``**config`` works as script.py expects.
"""
from typing import Any, Dict, Optional

import yaml


class ConfigParser(dict):
    def __init__(self, defaults: Optional[Dict[str, Any]] = None) -> None:
        super().__init__()
        self.update(defaults or {})

    def read(self, config_path: str, verbose: bool = False) -> None:
        with open(config_path, 'rb') as f:
            self.update(yaml.safe_load(f))
        if verbose:
            for k in self:
                print(f"{k}: {self[k]}")

    def get(self, item: str) -> Any:          # noqa: A003  (KeyError on a missing key, as the reference)
        return self[item]
