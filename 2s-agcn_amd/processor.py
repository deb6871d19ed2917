"""``Processor``: counterpart of the reference training runtime for THIS hot path only
(reference utils/processor.py: load_model :286-343, adjust_learning_rate :349-360, train :604-778, eval :784-914,
save/load_weights :225-270) and of its config surface (utils/parser.py:9-282).

What is kept: the flag names and yaml keys of the AGCN configs (``model``, ``model_args``, ``base_lr``, ``step``,
``weight_decay``, ``nesterov``, ``batch_size``, ``test_batch_size``, ``num_epoch``, ``warm_up_epoch``, ``device``,
``weights``, ``ignore_weights``, ``work_dir`` ...), precedence defaults < config file < command line, dotted class
paths resolved by ``import_class``, state_dict checkpoints named ``<work_dir>/weight/<model>-<epoch>-<step>.pt``,
per-epoch LR rule, mean-CE loss, clip 1.0, SGD.  What differs (MI355X-first): one process per GPU, flat
parameter/gradient buffers, one RCCL all-reduce and one fused clip+SGD launch chain per step (trainer.py), tensor
or tuple model outputs both accepted, unknown config keys warn instead of assert, and a
built-in synthetic feeder so the loop runs without a dataset.  Out of scope here: SAM/Adam/LLRD optimizers,
TensorBoard, the SGN feeders.
"""
import argparse
import importlib
import os
import pickle
import time

import numpy as np
import torch
import yaml

from . import dp as _dp
from .trainer import TrainEngine, learning_rate


def import_class(name):
    module_name, _, cls = name.rpartition('.')
    return getattr(importlib.import_module(module_name), cls)


def str2bool(v):
    return str(v).lower() in ('yes', 'true', 't', 'y', '1')


def get_parser():
    p = argparse.ArgumentParser(description='2s-AGCN on MI355X')
    p.add_argument('--work-dir', default='./work_dir/temp')
    p.add_argument('--config', default=None)
    p.add_argument('--phase', default='train')
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--log-interval', type=int, default=100)
    p.add_argument('--save-interval', type=int, default=2)
    p.add_argument('--eval-interval', type=int, default=5)
    p.add_argument('--print-log', type=str2bool, default=True)
    p.add_argument('--show-topk', type=int, default=[1, 5], nargs='+')
    p.add_argument('--feeder', default='agcn_amd.processor.SyntheticFeeder')
    p.add_argument('--num-worker', type=int, default=0)
    p.add_argument('--train-feeder-args', default=dict())
    p.add_argument('--test-feeder-args', default=dict())
    p.add_argument('--model', default='model.agcn.Model')
    p.add_argument('--model-args', default=dict())
    p.add_argument('--model-saved-name', default='')
    p.add_argument('--weights', default=None)
    p.add_argument('--ignore-weights', type=str, default=[], nargs='+')
    p.add_argument('--base-lr', type=float, default=0.1)
    p.add_argument('--step', type=int, default=[30, 40], nargs='+')
    p.add_argument('--device', type=int, default=0, nargs='+')
    p.add_argument('--optimizer', default='SGD')
    p.add_argument('--nesterov', type=str2bool, default=True)
    p.add_argument('--batch-size', type=int, default=64)
    p.add_argument('--test-batch-size', type=int, default=64)
    p.add_argument('--start-epoch', type=int, default=0)
    p.add_argument('--num-epoch', type=int, default=50)
    p.add_argument('--weight-decay', type=float, default=1e-4)
    p.add_argument('--warm-up-epoch', type=int, default=0)
    p.add_argument('--only-train-part', type=str2bool, default=False)
    p.add_argument('--only-train-epoch', type=int, default=0)
    p.add_argument('--ddp', type=str2bool, default=False)
    p.add_argument('--sync-bn', type=str2bool, default=None,
                   help='synchronise BatchNorm statistics over the ranks.  Default: on whenever more than one process '
                        'trains (the reference\'s only multi-process path always converts to SyncBatchNorm, '
                        'processor.py:295); "--sync-bn false" opts out to per-replica statistics (the semantics of '
                        'the reference\'s single-process nn.DataParallel path, processor.py:336-343)')
    p.add_argument('--world-size', type=int, default=1)
    p.add_argument('--max-steps-per-epoch', type=int, default=0, help='0 = whole epoch (synthetic smoke runs)')
    p.add_argument('--save-score', type=str2bool, default=False,
                   help='write <work_dir>/score/epoch<N>_<loader>.pkl {sample_name: logits} after an eval '
                        '(reference processor.py:199-210), the input of ensemble.py')
    p.add_argument('--device-augment', type=str2bool, default=True,
                   help='apply the feeder\'s random transforms to whole batches on the GPU (feeders.DeviceAugment) '
                        'instead of per sample on the host; only feeders that support it (feeders.feeder.Feeder)')
    p.add_argument('--prefetch-depth', type=int, default=2, help='batches staged ahead through pinned host buffers')
    return p


def load_args(argv=None):
    """defaults < yaml config < command line (reference utils/parser.py:248-282); unknown config keys only warn."""
    parser = get_parser()
    pre, _ = parser.parse_known_args(argv)
    if pre.config is not None:
        with open(pre.config, 'r') as f:
            cfg = yaml.safe_load(f) or {}
        known = vars(pre).keys()
        unknown = [k for k in cfg if k not in known]
        if unknown:
            print(f'[agcn_amd] ignoring config keys outside the AGCN hot path: {unknown}')
        parser.set_defaults(**{k: v for k, v in cfg.items() if k in known})
    return parser.parse_args(argv)


class SyntheticFeeder(torch.utils.data.Dataset):
    """Stand-in for the reference Feeder (feeders/feeder.py:35-227) when no dataset is present: seeded N(0,1) clips of
    shape (3, T, V, M) with uniform labels; __getitem__ returns (data, label, index) like the reference."""

    def __init__(self, num_samples=256, num_point=25, num_class=60, window_size=300, num_person=2, seed=0, **_):
        g = np.random.default_rng(seed)
        self.data = g.standard_normal((num_samples, 3, window_size, num_point, num_person)).astype(np.float32)
        self.label = g.integers(0, num_class, size=(num_samples,)).astype(np.int64)

    def __len__(self):
        return len(self.label)

    def __getitem__(self, i):
        return self.data[i], self.label[i], i


class NpyFeeder(torch.utils.data.Dataset):
    """Reads the reference's on-disk format (data_gen/ntu_gendata.py:158-173): ``data_path`` .npy (N,3,T,V,M) fp32
    (memory-mapped) and ``label_path`` .pkl (names, labels) or .npy labels."""

    def __init__(self, data_path, label_path, **_):
        self.data = np.load(data_path, mmap_mode='r')
        if label_path.endswith('.npy'):
            self.label = np.load(label_path).astype(np.int64)
        else:
            with open(label_path, 'rb') as f:      # the user's own dataset file, as in the reference feeder
                _, labels = pickle.load(f, encoding='latin1')
            self.label = np.asarray(labels, dtype=np.int64)

    def __len__(self):
        return len(self.label)

    def __getitem__(self, i):
        return np.asarray(self.data[i], dtype=np.float32), self.label[i], i


def use_device_augment(Feeder, feeder_kwargs, flag):
    """Whether the feeder's transforms run batch-wise on the GPU (feeders.DeviceAugment) for this configuration.
    DeviceAugment implements neither ``stretch`` nor ``random_subsample`` (reference feeders/feeder.py:184-185, 206-208):
    a config that asks for one of them keeps ALL transforms on the host, in the reference's order, instead of silently
    training on different data."""
    if not (bool(flag) and hasattr(Feeder, 'rotation_theta')):
        return False
    return not (feeder_kwargs.get('stretch') or feeder_kwargs.get('random_subsample') is not None)


def prepare_batch(data, aug):
    """What BOTH phases feed the model: the batch as float32 with the feeder's (device-side) transforms applied.  The
    reference applies normalization / window padding / ... in ``__getitem__`` whatever the phase (feeder.py:182-221);
    with device_augment the feeder returns raw clips and the same transforms live in ``aug``."""
    data = data.float()
    return aug(data) if aug is not None else data


class _IndexSampler(torch.utils.data.Sampler):
    """Yields the index list it currently holds (set per epoch by ``Processor._loader``)."""

    def __init__(self):
        self.indices = []

    def __iter__(self):
        return iter(list(self.indices))

    def __len__(self):
        return len(self.indices)


class Processor:
    def __init__(self, arg):
        self.arg = arg
        self.rank, self.world = _dp.init_distributed()
        dev = arg.device if isinstance(arg.device, int) else arg.device[0]
        local = int(os.environ.get('LOCAL_RANK', dev))
        torch.cuda.set_device(local)
        self.device = torch.device('cuda', local)
        torch.manual_seed(arg.seed)
        np.random.seed(arg.seed)
        os.makedirs(os.path.join(arg.work_dir, 'weight'), exist_ok=True)
        self.global_step = 0
        if str(arg.optimizer).upper() != 'SGD':
            raise ValueError(f"agcn_amd.Processor: optimizer {arg.optimizer!r} is not built for this path (SGD with "
                             f"momentum/nesterov only: reference processor.py:395-401); SAM/Adam/LLRD are out of scope")
        self.load_model()
        self.load_data()
        sync_bn = getattr(arg, 'sync_bn', None)
        self.sync_bn = (self.world > 1) if sync_bn is None else (bool(sync_bn) and self.world > 1)
        if self.sync_bn:
            self.model = _dp.enable_sync_bn(self.model, self.world)
        self.engine = TrainEngine(self.model, base_lr=arg.base_lr, momentum=0.9, nesterov=arg.nesterov,
                                  weight_decay=arg.weight_decay, max_grad_norm=1.0, world_size=self.world)
        _dp.broadcast_parameters(self.engine.fp.flat, self.world)
        self.best_acc = 0.0

    # ---- logging --------------------------------------------------------------------------------------------------
    def print_log(self, msg):
        if self.rank == 0:
            line = f"[ {time.strftime('%a %b %d %H:%M:%S %Y')} ] {msg}"
            print(line, flush=True)
            if self.arg.print_log:
                with open(os.path.join(self.arg.work_dir, 'log.txt'), 'a') as f:
                    print(line, file=f)

    # ---- model / weights -----------------------------------------------------------------------------------------
    def load_model(self):
        Model = import_class(self.arg.model)
        self.model = Model(**self.arg.model_args).to(self.device)
        if self.arg.weights:
            self.load_weights(self.arg.weights)

    def load_weights(self, path):
        name = os.path.basename(path)
        try:
            self.global_step = int(name[:-3].split('-')[-1])     # <model>-<epoch>-<global_step>.pt
        except ValueError:
            self.global_step = 0
        weights = torch.load(path, map_location='cpu', weights_only=True)
        weights = {k.replace('module.', '', 1) if k.startswith('module.') else k: v for k, v in weights.items()}
        for key in self.arg.ignore_weights:
            for k in [k for k in weights if key in k]:
                weights.pop(k)
                self.print_log(f'Successfully removed weights: {k}')
        state = self.model.state_dict()
        missing = set(state) - set(weights)
        if missing:
            self.print_log(f'Cannot find these weights (kept at init): {sorted(missing)}')
        state.update({k: v for k, v in weights.items() if k in state})
        self.model.load_state_dict(state)

    def save_weights(self, epoch):
        if self.rank != 0:
            return None
        sd = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        # reference: <work_dir>/weight/<ModelClass>-<epoch>-<global_step>.pt (processor.py:225-231)
        name = self.arg.model_saved_name or os.path.join(self.arg.work_dir, 'weight', self.arg.model.split('.')[-1])
        path = f'{name}-{epoch}-{int(self.global_step)}.pt'
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        torch.save(sd, path)
        return path

    # ---- data -----------------------------------------------------------------------------------------------------
    def load_data(self):
        """Datasets + per-rank sampling (reference processor.py:479-520 / feeders/loader.py:378-393).  Batches reach the
        GPU through a pinned double-buffered ring (feeders.DeviceLoader); feeders that can defer their random
        transforms get them applied batch-wise on the device (feeders.DeviceAugment)."""
        from .feeders import DeviceAugment
        Feeder = import_class(self.arg.feeder)
        self.datasets, self.augment, self._loaders = {}, {}, {}

        def build(kwargs, train):
            kw = dict(kwargs)
            dev_aug = use_device_augment(Feeder, kw, self.arg.device_augment)
            if dev_aug:
                kw['device_augment'] = True
            ds = Feeder(**kw)
            return ds, (DeviceAugment.from_feeder(ds) if dev_aug else None)
        if self.arg.phase == 'train':
            self.datasets['train'], self.augment['train'] = build(self.arg.train_feeder_args, True)
        self.datasets['test'], self.augment['test'] = build(self.arg.test_feeder_args or self.arg.train_feeder_args, False)

    def _loader(self, name, epoch):
        """One epoch of device batches for this rank.  Train: DistributedSampler semantics -- every rank shuffles the
        SAME seeded permutation (seed + epoch, ``set_epoch``), pads it by wrap-around to a multiple of the world size and
        takes every world-th index.  Test: strided shards in order, the padded tail is dropped again when the scores
        are gathered (reference processor.py:862-868)."""
        from .feeders import DeviceLoader
        ds = self.datasets[name]
        n = len(ds)
        if name == 'train':
            g = torch.Generator().manual_seed(int(self.arg.seed) + int(epoch))
            order = torch.randperm(n, generator=g).tolist()
            bs, drop = self.arg.batch_size, True
        else:
            order, bs, drop = list(range(n)), self.arg.test_batch_size, False
        per = (n + self.world - 1) // self.world
        order = order + order[:per * self.world - n]
        idx = order[self.rank::self.world]
        # ONE DataLoader per split for the whole run (persistent workers: forking and warming them per epoch would
        # cost seconds); the epoch's index list is swapped into its sampler
        if name not in self._loaders:
            sampler = _IndexSampler()
            nw = int(self.arg.num_worker)
            dl = torch.utils.data.DataLoader(ds, batch_size=bs, sampler=sampler, drop_last=drop, num_workers=nw,
                                             persistent_workers=nw > 0, prefetch_factor=4 if nw > 0 else None)
            self._loaders[name] = (dl, sampler)
        dl, sampler = self._loaders[name]
        sampler.indices = idx
        return DeviceLoader(dl, self.device, depth=self.arg.prefetch_depth), idx

    # ---- one epoch ------------------------------------------------------------------------------------------------
    def train(self, epoch):
        self.engine.lr = learning_rate(epoch, self.arg.base_lr, self.arg.step, self.arg.warm_up_epoch)
        freeze_pa = self.arg.only_train_part and epoch <= self.arg.only_train_epoch
        self.print_log(f'Training epoch: {epoch + 1}, lr {self.engine.lr:.6f}')
        # the reference's three buckets (processor.py:608, 670, 709, 752, 759-775): waiting for data / the step /
        # statistics + logging, as host wall time like the reference.  Nothing synchronises the device inside the loop
        # (the step is asynchronous; the host runs ahead of the GPU), so `dataloader` is the time the training thread
        # actually WAITED for its next batch; the device is drained once at the end of the epoch and that wait is
        # booked on `model`.
        timer = dict(dataloader=0.0, model=0.0, statistics=0.0)
        losses = []
        loader, _ = self._loader('train', epoch)
        aug = self.augment.get('train')
        mark = time.perf_counter()

        def split():
            nonlocal mark
            now = time.perf_counter()
            dt, mark = now - mark, now
            return dt
        for step, (data, label, _) in enumerate(loader):
            timer['dataloader'] += split()
            data = prepare_batch(data, aug)
            if freeze_pa:
                loss = self._train_step_frozen_pa(data, label)
            else:
                loss = self.engine.train_step(data, label)
            self.global_step += 1
            timer['model'] += split()
            if step % self.arg.log_interval == 0:
                losses.append(float(loss.detach()))
                self.print_log(f'\tBatch({step}/{len(loader)}) done. Loss: {losses[-1]:.4f}  '
                               f'lr:{self.engine.lr:.6f}  grad-norm:{self.engine.grad_norm():.3f}')
            timer['statistics'] += split()
            if self.arg.max_steps_per_epoch and step + 1 >= self.arg.max_steps_per_epoch:
                break
        torch.cuda.current_stream().synchronize()
        timer['model'] += split()
        tot = max(sum(timer.values()), 1e-9)
        self.last_timer = dict(timer)
        self.print_log(f'\tMean training loss: {np.mean(losses) if losses else float("nan"):.4f}.')
        self.print_log(f'\tTime consumption  : [Data] {100 * timer["dataloader"] / tot:02.0f}%, '
                       f'[Network] {100 * timer["model"] / tot:02.0f}%, [Statistics] {100 * timer["statistics"] / tot:02.0f}%')

    def _train_step_frozen_pa(self, data, label):
        """Reference order while ``epoch <= only_train_epoch`` (processor.py:697-703): backward, clip_grad_norm_ over ALL
        gradients (PA included), THEN zero the PA gradients, then optimizer.step() -- so PA still takes the weight-decay /
        momentum part of the SGD update.  The fused clip+SGD kernel runs on the true gradients; the PA slices are then
        recomputed as the step a zero gradient would have produced."""
        eng = self.engine
        slices = []
        for (n, p), o in zip([(n, p) for n, p in self.model.named_parameters() if p.requires_grad], eng.fp.offsets):
            if 'PA' in n:
                slices.append((o, p.numel()))
        saved = [(eng.fp.flat[o:o + k].clone(), eng.fp.momentum[o:o + k].clone()) for o, k in slices]
        first = eng.steps_done == 0
        loss = eng.train_step(data, label)
        mu, wd, lr = eng.momentum, eng.weight_decay, eng.lr
        for (o, k), (p0, m0) in zip(slices, saved):
            d = wd * p0
            m1 = d.clone() if first else mu * m0 + d
            upd = d + mu * m1 if eng.nesterov else m1
            eng.fp.flat[o:o + k] = p0 - lr * upd
            eng.fp.momentum[o:o + k] = m1
        return loss

    def eval(self, epoch, save_score=None, loader_name='test'):
        """reference processor.py:784-914: eval-mode forward over the shard, scores gathered over the ranks and
        de-interleaved (:862-868), top-k through the dataset's labels, optional score pickle (:199-210)."""
        self.model.eval()
        loader, idx = self._loader(loader_name, epoch)
        ds = self.datasets[loader_name]
        # the feeder's transforms run in BOTH phases (reference feeder.py:182-221: normalization, window_size padding
        # ... are applied by __getitem__ whatever the phase); with device_augment they live in DeviceAugment
        aug = self.augment.get(loader_name)
        scores, loss_sum, seen = [], 0.0, 0
        with torch.no_grad():
            for data, label, _ in loader:
                out = self.model(prepare_batch(data, aug))
                out = out[0] if isinstance(out, tuple) else out
                loss_sum += float(torch.nn.functional.cross_entropy(out, label, reduction='sum'))
                seen += label.numel()
                scores.append(out.float().cpu().numpy())
        score = np.concatenate(scores) if scores else np.zeros((0, 1), dtype=np.float32)
        n = len(ds)
        if self.world > 1:
            import torch.distributed as dist
            parts = [None] * self.world
            dist.all_gather_object(parts, score)
            full = np.zeros((len(parts[0]) * self.world, score.shape[1]), dtype=score.dtype)
            for r, val in enumerate(parts):
                full[r::self.world] = val               # rank r holds samples r, r + W, ... (reference :866-868)
            score = full[:n]                            # the wrap-around padding of the shards is dropped again
            loss_sum = _dp.allreduce_scalar(loss_sum, self.world, self.device)
            seen = _dp.allreduce_scalar(seen, self.world, self.device)
        labels = np.asarray(ds.label)[:n]
        res = {}
        for k in self.arg.show_topk:
            kk = min(k, score.shape[1])
            res[k] = float(np.mean([labels[i] in np.argsort(score[i])[-kk:] for i in range(n)])) if n else 0.0
        loss = loss_sum / max(seen, 1)
        self.print_log(f'\tMean {loader_name} loss of {n} samples: {loss:.4f}')
        for k, acc in res.items():
            self.print_log(f'\tTop{k}: {100.0 * acc:.2f}%')
        self.best_acc = max(self.best_acc, res.get(1, 0.0))
        if (self.arg.save_score if save_score is None else save_score) and self.rank == 0:
            self.save_scores(epoch, loader_name, score)
        self.last_score = score
        return loss, res

    def save_scores(self, epoch, loader_name, score):
        """{sample_name: logits} pickle, the file ``ensemble.py`` fuses (reference processor.py:199-210)."""
        names = getattr(self.datasets[loader_name], 'sample_name', None)
        names = list(names) if names is not None else list(range(len(score)))
        path = os.path.join(self.arg.work_dir, 'score', f'epoch{epoch + 1}_{loader_name}.pkl')
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'wb') as f:
            pickle.dump(dict(zip(names, score)), f)
        return path

    def start(self):
        if self.arg.phase == 'train':
            self.print_log(f'Parameters: {sum(p.numel() for p in self.model.parameters())}, world {self.world}')
            for epoch in range(self.arg.start_epoch, self.arg.num_epoch):
                self.train(epoch)
                if (epoch + 1) % self.arg.save_interval == 0 or epoch + 1 == self.arg.num_epoch:
                    self.save_weights(epoch + 1)
                if (epoch + 1) % self.arg.eval_interval == 0 or epoch + 1 == self.arg.num_epoch:
                    self.eval(epoch)
            self.print_log(f'best accuracy: {self.best_acc}')
        else:
            if self.arg.weights is None:
                raise ValueError('Please appoint --weights.')
            self.eval(0)
