"""Per-stage SHA-256 digests of a seeded forward + backward of three fixed fixtures (AGCN m_ntu_b1, AAGCN am_ntu_l3_t32 and
am_ntu_b1_t64), one line per stage, so that the logs of any two processes / boxes can be compared with `diff`: the first
line that differs names the first stage whose bits differ.

    python tools/stage_checksums.py [tag] > gpurun_out/checksums_<tag>.txt

Stages: the prologue (data_bn output reshaped to the unit layout), every TCN_GCN_unit output, the pooled features, the
logits, then (backward) the gradient of every unit input and every parameter gradient.  The final line `ALL` is the digest
of (logits, flat gradient) per fixture -- the value tests/golden/determinism.json pins."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import agcn_oracle as orc  # noqa: E402   (parameter recipes of the fixtures only)
from tests import golden_util as gu  # noqa: E402


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def build(name, dev):
    gold = gu.load(name)
    n, v, num_class, seed, t = [int(i) for i in gold['meta']]
    if name.startswith('am_'):
        from model.aagcn import Model
        layers, gbn = gu.meta_int(gold, 'meta.layers', 10), gu.meta_int(gold, 'meta.gbn') or None
        model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.ntu_rgb_d.Graph',
                      graph_args=dict(labeling_mode='spatial'), model_layers=layers, gbn_split=gbn)
        shapes = orc.aagcn_model_param_shapes(num_class, v, model_layers=layers, gbn_split=gbn)
        model.load_state_dict(orc.aagcn_randomized_state(shapes, seed, stress=float(gold['meta.stress'])))
    else:
        from model.agcn import Model
        graph = 'graph.ntu_rgb_d.Graph' if v == 25 else 'graph.kinetics.Graph'
        model = Model(num_class=num_class, num_point=v, num_person=2, graph=graph,
                      graph_args=dict(labeling_mode='spatial'))
        shapes = orc.model_param_shapes(num_class, v)
        model.load_state_dict(orc.randomized_state(shapes, seed, stress=float(gold['meta.stress'])))
    xn, lab = gu.model_inputs(n, v, num_class, seed, t)
    return model.to(dev).train(), torch.from_numpy(xn).to(dev), torch.from_numpy(lab).to(dev)


def detail_hooks(model, name, out):
    """DETAIL=1: also hash the sub-stages of every unit (GCN core output before the attention gates, the gate tensors,
    the unit_gcn / unit_tcn operator outputs), in execution order."""
    from agcn_amd import ops
    seq = [0]

    def note(tag, t):
        if torch.is_tensor(t):
            out.append(f'{name} det {seq[0]:04d} {tag:28s} {sha(t)}')
            seq[0] += 1
    orig_gates = ops.STCAttentionFunction._gates

    def gates(m_s, mv1, *a, **k):
        note('gates.in.m_s', m_s)
        note('gates.in.mv1', mv1)
        r = orig_gates(m_s, mv1, *a, **k)
        for tag, t in zip(('a_s', 'a_t', 'a_c'), r):
            note('gates.out.' + tag, t)
        return r
    ops.STCAttentionFunction._gates = staticmethod(gates)
    for fn in ('adjacency_fused_fwd', 'aggregate_project_fwd', 'gcn_first_fwd', 'conv_fwd', 'bn_act_fwd', 'stc_row_reduce'):
        orig = getattr(ops, fn)

        def wrap(*a, _o=orig, _f=fn, **k):
            r = _o(*a, **k)
            for j, t in enumerate(r if isinstance(r, tuple) else (r,)):
                note(f'{_f}[{j}]', t)
            return r
        setattr(ops, fn, wrap)


def run(name, dev, out):
    model, x, lab = build(name, dev)
    cap = {}
    if os.environ.get('DETAIL'):
        detail_hooks(model, name, out)
    units = [(k, m) for k, m in model.named_children() if k.startswith('l') and k[1:].isdigit()]

    def pre(k):
        def hook(mod, inp):
            if inp[0].requires_grad:
                inp[0].retain_grad()
            cap['in.' + k] = inp[0]
        return hook

    def post(k):
        def hook(mod, inp, o):
            cap['out.' + k] = o
        return hook
    for k, m in units:
        m.register_forward_pre_hook(pre(k))
        m.register_forward_hook(post(k))
    if hasattr(model, 'fc'):
        model.fc.register_forward_pre_hook(lambda mod, inp: cap.__setitem__('pooled', inp[0]))
    res = model(x)
    logits = res[0] if isinstance(res, tuple) else res
    loss = torch.nn.functional.cross_entropy(logits, lab)
    loss.backward()
    torch.cuda.synchronize()
    out.append(f'{name} fwd prologue      {sha(cap["in." + units[0][0]])}')
    for k, _ in units:
        out.append(f'{name} fwd out.{k:4s}      {sha(cap["out." + k])}')
    if 'pooled' in cap:
        out.append(f'{name} fwd pooled        {sha(cap["pooled"])}')
    out.append(f'{name} fwd logits        {sha(logits)}')
    out.append(f'{name} fwd loss          {sha(loss)}')
    for k, _ in reversed(units):
        g = cap['in.' + k].grad
        if g is not None:
            out.append(f'{name} bwd din.{k:4s}      {sha(g)}')
    flat = []
    for pn, p in model.named_parameters():
        if p.grad is not None:
            out.append(f'{name} bwd g.{pn:40s} {sha(p.grad)}')
            flat.append(p.grad.detach().flatten())
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(logits.detach().cpu().numpy()).tobytes())
    h.update(np.ascontiguousarray(torch.cat(flat).cpu().numpy()).tobytes())
    out.append(f'{name} ALL {h.hexdigest()}')
    return h.hexdigest()


FIXTURES = ['m_ntu_b1', 'am_ntu_l3_t32', 'am_ntu_b1_t64']


def csrc_fingerprint():
    """sha256 over the kernel sources and the host operator chain: the build a set of pinned digests is valid for."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, '2s-agcn_amd', 'csrc', '*.hip')) +
                   glob.glob(os.path.join(root, '2s-agcn_amd', 'csrc', '*.h')) +
                   [os.path.join(root, '2s-agcn_amd', 'ops.py'), os.path.join(root, '2s-agcn_amd', 'model', 'agcn.py'),
                    os.path.join(root, '2s-agcn_amd', 'model', 'aagcn.py')])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    import agcn_amd  # noqa: F401
    dev = torch.device('cuda:0')
    args = sys.argv[1:]
    write = None
    if '--write' in args:
        write = args[args.index('--write') + 1]
    names = [a for a in args if a in FIXTURES] or FIXTURES
    lines, digests = [], {}
    for name in names:
        digests[name] = run(name, dev, lines)
    print('\n'.join(lines))
    if write:
        import json
        with open(write, 'w') as f:
            json.dump({'_meta': {'csrc_sha': csrc_fingerprint(), 'device': torch.cuda.get_device_name(0),
                                 'torch': torch.__version__,
                                 'what': 'sha256(logits bytes + flat parameter-gradient bytes) of one seeded training '
                                         'forward+backward per fixture (tools/stage_checksums.py)'},
                       'digests': digests}, f, indent=1)


if __name__ == '__main__':
    main()
