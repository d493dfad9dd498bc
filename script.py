"""
Entrypoint: ``python -m script --mode {train,eval} --models_dir models_dir --run_name <run> --data_dir <dir>``.

Mirrors /root/reference/script.py: same four CLI flags (:20-29), same config surface (``models_dir/<run>/config.yaml``
merged over ``{mode, data_dir, checkpoint_dir, log_dir}``, :32-46), one process per rank via ``mp.spawn`` (:129-136),
``init_process_group(backend, world_size, rank)`` with ``master_addr/master_port`` from the config (:50-56).  What
changes: on an MI355X node the backend is RCCL over xGMI whatever the YAML says (``nccl`` IS RCCL under PyTorch-ROCm;
``gloo`` is kept for CPU-only plumbing runs, which cannot execute the HIP path), each rank binds to its own GPU, and
DistributedDataParallel (:64-71) is replaced by ``ddp.GradReducer`` (bucketed all-reduce overlapped with the engine's
backward) + a rank-0 parameter broadcast.  ``--data_dir synthetic`` feeds the fixed-shape synthetic batches the benchmark uses; any
other ``--data_dir`` must hold the extracted CIFAR archive (as torchvision leaves it): the dataset then lives in HBM and the ``data_aug``
chains run as one launch per batch (``utils/data_util.py``).
"""
import argparse
import os

import torch

from pytorch_ddp_resnet_amd import ResNet
from pytorch_ddp_resnet_amd.algos.evaluation import evaluation_loop
from pytorch_ddp_resnet_amd.algos.training import training_loop
from pytorch_ddp_resnet_amd.ddp import GradReducer, broadcast_parameters
from pytorch_ddp_resnet_amd.utils.amp import GradScaler
from pytorch_ddp_resnet_amd.utils.checkpoint_util import ddp_keys, get_checkpoint_strategy, maybe_load_checkpoints
from pytorch_ddp_resnet_amd.utils.config_util import ConfigParser
from pytorch_ddp_resnet_amd.utils.optim_util import get_optimizer, get_scheduler

_SHAPES = {'CIFAR10': (32, 10), 'CIFAR100': (32, 100), 'ImageNet': (224, 1000)}


def create_argparser():
    parser = argparse.ArgumentParser(description="MI355X-native ResNet trainer (drop-in for lucaslingle/pytorch_ddp_resnet's script.py)")
    parser.add_argument("--mode", choices=['train', 'eval'], default='train')
    parser.add_argument("--models_dir", type=str, default='models_dir')
    parser.add_argument("--run_name", type=str, default='wrn-28-10-dropout_cifar10')
    parser.add_argument("--data_dir", type=str, default='synthetic')
    return parser


def get_config(args):
    base = os.path.join(args.models_dir, args.run_name)
    config = ConfigParser(defaults={'mode': args.mode, 'data_dir': args.data_dir, 'checkpoint_dir': os.path.join(base, 'checkpoints'),
                                    'log_dir': os.path.join(base, 'tensorboard_logs')})
    config.read(os.path.join(base, 'config.yaml'), verbose=True)
    return config


class SyntheticLoader:
    """fixed-shape N(0,1) NCHW fp32 batches resident on the device: `steps` microbatches per epoch."""

    def __init__(self, batch, hw, classes, steps, device, seed):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(batch, 3, hw, hw, generator=g).to(device)
        self.y = torch.randint(0, classes, (batch,), generator=g).to(device)
        self.steps = steps

    def __iter__(self):
        for _ in range(self.steps):
            yield self.x, self.y


def setup(rank, config):
    os.environ['MASTER_ADDR'] = config.get('master_addr')
    os.environ['MASTER_PORT'] = str(config.get('master_port'))
    on_gpu = torch.cuda.is_available()
    backend = 'nccl' if on_gpu else config.get('backend')
    world = config.get('world_size')
    if on_gpu:
        torch.cuda.set_device(rank)
    torch.distributed.init_process_group(backend=backend, world_size=world, rank=rank)
    device = torch.device('cuda', rank) if on_gpu else torch.device('cpu')
    sampler_train = None
    if config.get('data_dir') != 'synthetic':
        # the dataset archive under data_dir, resident on the device; sampler = the reference's DistributedSampler(seed=0); the
        # data_aug chains as one launch per batch (utils/data_util.py, utils/transform_util.py)
        from pytorch_ddp_resnet_amd.utils import data_util
        if not on_gpu:
            raise RuntimeError("the input pipeline runs on the device (rn_augment_batch): no GPU, no CPU fallback")
        datasets = data_util.get_datasets(device=device, **config)
        samplers = data_util.get_samplers(rank, **config, **datasets)
        loaders = data_util.get_dataloaders(rank=rank, **config, **datasets, **samplers)
        dl_train, dl_test, sampler_train = loaders['dl_train'], loaders['dl_test'], samplers['sampler_train']
    else:
        hw, classes = _SHAPES[config.get('dataset_cls_name')]
        local_batch = config.get('batch_size') // (config.get('num_microbatches') * world)          # data_util.py:216 (batch_size is global)
        steps = config.get('num_microbatches') * 4
        dl_train = SyntheticLoader(local_batch, hw, classes, steps, device, 1234 + rank)
        dl_test = SyntheticLoader(local_batch, hw, classes, 2, device, 4321 + rank)
    # the reference turns AMP (fp16 autocast + GradScaler) on whenever a GPU is present (script.py:63, training.py:95-110): the
    # engine's counterpart is fp16 storage / f16 MFMA with fp32 accumulation, statistics and master weights, under the same
    # GradScaler.  `compute_dtype: fp32 | bf16 | fp16` in the YAML overrides it (neither fp32 nor bf16 needs a scaler).
    compute_dtype = dict(config).get('compute_dtype', 'fp16')
    scaler = GradScaler('cuda') if (on_gpu and compute_dtype == 'fp16') else None      # torch.amp.GradScaler, inspecting the flat gradient buffer
    classifier = ResNet(architecture_spec=config.get('architecture_spec'), preact=config.get('preact'), use_proj=config.get('use_proj'),
                        dropout_prob=config.get('dropout_prob'), sync_bn=bool(dict(config).get('sync_bn', False)),
                        compute_dtype=compute_dtype).to(device)
    if world > 1:
        broadcast_parameters(classifier)                       # ranks draw different initial weights (no seeding): rank 0 defines the model
    reducer = GradReducer(classifier, world) if world > 1 else None
    optimizer = get_optimizer(config.get('optimizer_cls_name'), classifier, config.get('optimizer_args'))
    scheduler = get_scheduler(config.get('scheduler_cls_name'), optimizer, config.get('scheduler_args'))
    # resume (/root/reference/script.py:80-94): every checkpointable from its newest `{kind}_{steps}.pth`; the classifier through DDP's `module.` key
    # scheme, so files written by the reference (or by this trainer) load into either.  All ranks read the same files.
    cfg = dict(config)
    checkpoint_strategy = None
    if cfg.get('checkpoint_strategy_cls_name'):
        checkpoint_strategy = get_checkpoint_strategy(cfg['checkpoint_strategy_cls_name'], cfg.get('checkpoint_strategy_args'))
    kinds = {'checkpoint_strategy': checkpoint_strategy, 'classifier': ddp_keys(classifier), 'optimizer': optimizer, 'scheduler': scheduler, 'scaler': scaler}
    if config.get('mode') != 'train':
        kinds = {'classifier': kinds['classifier']}         # evaluation reads the weights only: a directory holding just classifier_{steps}.pth is enough
    global_step = maybe_load_checkpoints(config.get('checkpoint_dir'), kinds, map_location=device, steps=None)
    return dict(device=device, dl_train=dl_train, dl_test=dl_test, classifier=classifier, optimizer=optimizer, scheduler=scheduler,
                reducer=reducer, global_step=global_step, scaler=scaler, sampler_train=sampler_train, checkpoint_strategy=checkpoint_strategy)


def train(rank, config):
    system = setup(rank, config)
    training_loop(rank, **{k: v for k, v in config.items() if k not in system}, **system)
    torch.distributed.destroy_process_group()


def evaluate(rank, config):
    system = setup(rank, config)
    if system['global_step'] == 0 and rank == 0:
        print("WARNING: no classifier checkpoint under " + str(config.get('checkpoint_dir')) + ": evaluating randomly initialised weights")
    metrics = evaluation_loop(config.get('world_size'), system['device'], system['dl_test'], system['classifier'])
    if rank == 0:
        print(f"Test metrics: {metrics}")
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    args = create_argparser().parse_args()
    config = get_config(args)
    torch.multiprocessing.spawn(train if config.get('mode') == 'train' else evaluate, args=(config,), nprocs=config.get('world_size'), join=True)
