"""Benchmark of the AGCN hot path: skeleton-clips/sec, forward+backward(+clip+SGD step), NTU (N,3,300,25,2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU; per-GPU batch fixed at 64 (BASELINE.json configs[1]/[2], weak scaling); gradients are summed by
two RCCL all-reduces of slices of the flat gradient buffer per step (the tail of the model overlapped with the
backward); with more than one rank BatchNorm statistics are synchronised like the reference's DDP path.  Rank 0 prints one JSON line.  Inputs are synthetic and
resident in HBM before the timed region.  The roofline object is measured live (HIP events on the launch stream)
for the dominant kernel; the cpu_baseline object times the CPU oracle (a port of the reference, never the product
path) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: exact-f32 matrix rate (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak


WORKLOADS = {
    # name: (model module, num_class, num_point, graph, default per-GPU batch, description)
    'ntu_agcn': ('agcn', 60, 25, 'graph.ntu_rgb_d.Graph', 64, 'AGCN joint-stream NTU-xview shape'),
    'ntu_aagcn': ('aagcn', 60, 25, 'graph.ntu_rgb_d.Graph', 64, 'AAGCN (attention) NTU-xsub shape, fp32'),
    # BASELINE configs[3]: plain bf16 MFMA operands (one product per fp32 product), fp32 accumulate / storage / master weights
    'ntu_aagcn_bf16': ('aagcn', 60, 25, 'graph.ntu_rgb_d.Graph', 64, 'AAGCN (attention) NTU-xsub shape, bf16 MFMA'),
    'kinetics_agcn': ('agcn', 400, 18, 'graph.kinetics.Graph', 128, 'AGCN Kinetics-Skeleton shape (N,3,300,18,2)'),
}


def build_model(workload='ntu_agcn'):
    import importlib
    import agcn_amd  # noqa: F401
    mod, num_class, num_point, graph, _, _ = WORKLOADS[workload]
    Model = importlib.import_module('model.' + mod).Model
    return Model(num_class=num_class, num_point=num_point, num_person=2, graph=graph,
                 graph_args=dict(labeling_mode='spatial'))


def randomize_like_training(model, seed):
    """Random-init weights of the architecture, moved off the degenerate initialisation (bn weight 1e-6
    and PA 1e-6 make the adaptive branch numerically invisible); timing does not depend on the values."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('gcn1.bn.weight'):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif name.endswith('gcn1.PA'):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif name.endswith('agcn.alpha'):
                p.fill_(0.5)


def csrc_fingerprint():
    """sha256 over the kernel sources (2s-agcn_amd/csrc/*.hip, *.h): the key the committed PMC traffic file is valid for."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, '2s-agcn_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(ROOT, '2s-agcn_amd', 'csrc', '*.h'))):
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/pmc_traffic.sh -> profiles/pmc_traffic.json,
    FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 read correction).  Counters cannot be collected inside a
    timed run, so the file carries the fingerprint of the kernel sources it was measured on; a stale file (any kernel
    source edited since) is refused and the traffic is reported as null."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None, 'no profiles/pmc_traffic.json'
    meta = pmc.get('_meta', {})
    if meta.get('csrc_sha') != csrc_fingerprint():
        return None, f"stale: measured on csrc {meta.get('csrc_sha')}, this build is {csrc_fingerprint()}"
    key = [k for k in pmc if k.startswith(kernel)]
    if not key:
        return None, 'kernel not in the PMC file'
    return pmc[key[0]]['hbm_bytes'], f"profiles/pmc_traffic.json (csrc {meta.get('csrc_sha')}, commit {meta.get('git_head')})"


def dominant_kernel_roofline(device, reps=10):
    """The operator with the largest share of the step (profiles/*_kernel_stats.txt: the temporal convolution's
    conv_pc_kernel + conv_gemm_bf16_kernel + wgrad9_bf16_kernel; the unit_gcn chain gcn_chain_kernel comes next and is
    covered per layer by `unit_gcn_fwd`): unit_tcn's 9x1 temporal convolution, timed as the forward at the l9/l10 shape (N'=128,
    C=Cout=256, T=75, V=25) with HIP events on the stream it is launched on.  Algorithmic work per launch:
    2*Cout*Cin*9 FLOP per output position x 128*75*25 positions = 283.1 GFLOP (fp32-equivalent) and 491.5 MB.
    Arithmetic and ceiling follow the mode the library reports (agcn_gemm_mode): bf16x6 = every fp32 product is 6 bf16
    MFMA products with fp32 accumulation -> ceiling = dense bf16 rate / 6 = 416.7 TFLOP/s fp32-equivalent; f32 =
    exact-f32 MFMA, 157.3.  `kernel` is what the launch actually enqueued (agcn_last_kernel), `max_rel_err_vs_fp64` is
    the timed launch's output checked against an fp64 CPU convolution on one sample (fp32-equivalence on THIS box)."""
    import agcn_amd  # noqa: F401
    from agcn_amd import ops
    L = ops._L()
    mode = L.agcn_gemm_mode().decode()
    N, C, T, V = 128, 256, 75, 25
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, C, T, V, generator=g).to(device)
    w = (torch.randn(C, C, 9, 1, generator=g) / 48.0).to(device)
    b = torch.zeros(C, device=device)
    for _ in range(2):
        y, _ = ops.conv_fwd(x, w, b, 1, want_stats=True)
    kernel = L.agcn_last_kernel().decode()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv_fwd(x, w, b, 1, want_stats=True)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps          # includes the ~3 us weight-pack launch that precedes every call
    ref = torch.nn.functional.conv2d(x[:1].double().cpu(), w.double().cpu(), b.double().cpu(), padding=(4, 0))
    err = float((y[:1].double().cpu() - ref).abs().max() / ref.abs().max())
    flops = 2.0 * C * C * 9 * T * V * N
    achieved = flops / (ms * 1e-3) / 1e12
    if mode == 'f32':
        peak, note = PEAK_FP32_MFMA_TFLOPS, 'exact-f32 MFMA'
    elif kernel.endswith(', true>'):
        # the temporal convolutions of the default (fp32-equivalent) mode run on two fp16 pieces per operand and three
        # products (last template argument F16 = true): priced at a third of the dense 16-bit MFMA rate
        peak = round(PEAK_BF16_MFMA_TFLOPS / 3, 1)
        note = ('3 fp16 MFMA products per fp32 product (two fp16 pieces per operand, range-scaled by the tensor max), '
                f'fp32 accumulate: peak = {PEAK_BF16_MFMA_TFLOPS:.0f}/3')
    else:
        products = {'bf16x6': 6, 'bf16x3': 3, 'bf16': 1}[mode]
        peak = round(PEAK_BF16_MFMA_TFLOPS / products, 1)
        note = (f'{products} bf16 MFMA product(s) per fp32 product, fp32 accumulate: peak = '
                f'{PEAK_BF16_MFMA_TFLOPS:.0f}/{products}')
    traffic, src = pmc_traffic(kernel)
    return {"bound": "mfma", "kernel": kernel, "what": "unit_tcn 9x1 conv forward, l9-l10 shape (N'=128, C=256, T=75, V=25)",
            "arithmetic": note, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": src,
            "ms_per_launch": round(ms, 4), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": 4.0 * 2 * N * C * T * V, "max_rel_err_vs_fp64": float(f"{err:.3g}")}


def chain_kernel_roofline(device, reps=10):
    """The aggregate+project chain (unit_gcn's `sum_i Wd_i (x . A^_i)`, reference agcn.py:103-105) at the l2-l4 shape
    (N'=128, C=Cout=64, T=300, V=25), forward with BatchNorm partials, timed with HIP events on the launch stream with the
    operand maximum supplied (as the model does).  Algorithmic work: aggregate 3*2*C*T*V*V + project
    3*2*C*Cout*T*V FLOP per sample = 32.8 GFLOP, 491.5 MB.  Peak: f16x3 = 2500/3 TFLOP/s fp32-equivalent; HBM 8 TB/s."""
    import agcn_amd  # noqa: F401
    from agcn_amd import ops
    L = ops._L()
    N, C, T, V = 128, 64, 300, 25
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, C, T, V, generator=g).to(device)
    adj = (0.2 * torch.randn(N, 3, V, V, generator=g)).to(device)
    w = (torch.randn(C, 3 * C, generator=g) / (3 * C) ** 0.5).to(device)
    b = torch.zeros(C, device=device)
    amax = x.abs().max().reshape(1).contiguous()
    for _ in range(2):
        ops.aggregate_project_fwd(x, adj, w, b, want_stats=True, x_amax=amax)
    kernel = L.agcn_last_kernel().decode()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.aggregate_project_fwd(x, adj, w, b, want_stats=True, x_amax=amax)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps          # includes the ~5 us weight-pack launch of every call
    flops = N * (3 * 2.0 * C * T * V * V + 3 * 2.0 * C * C * T * V)
    nbytes = 4.0 * 2 * N * C * T * V
    chain_mode = L.agcn_chain_mode().decode()
    peak = round(PEAK_BF16_MFMA_TFLOPS / 3, 1) if chain_mode == 'f16x3' else PEAK_FP32_MFMA_TFLOPS
    t_mfma, t_hbm = flops / (peak * 1e12) * 1e3, nbytes / 8e12 * 1e3
    return {"kernel": kernel, "what": "aggregate+project chain forward, l2-l4 shape (N'=128, C=64, T=300, V=25)",
            "ms_per_launch": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 2), "GBps": round(nbytes / ms / 1e6, 1),
            "bound": "hbm" if t_hbm > t_mfma else "mfma", "frac": round(max(t_mfma, t_hbm) / ms, 4),
            "frac_of_hbm": round(t_hbm / ms, 4), "frac_of_mfma_f16x3": round(t_mfma / ms, 4),
            "flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes}


def exact_f32_value(args):
    """clips/s of the SAME step with every contraction on exact-f32 MFMA (AGCN_GEMM=f32), for context next to the
    split-arithmetic headline: AGCN_GEMM is read once per process, so this is a child process."""
    import subprocess
    env = dict(os.environ, AGCN_GEMM='f32')
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-roofline',
           '--workload', args.workload, '--batch', str(args.batch)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        return json.loads(r.stdout.strip().splitlines()[-1])['value']
    except Exception as ex:     # noqa: BLE001 -- context only; never fails the benchmark line
        print(f"[bench] exact-f32 context run failed: {ex}", file=sys.stderr, flush=True)
        return None


def top_profiled_kernel():
    """The kernel with the largest share of the step in the committed rocprofv3 profile of this round (profiles/r03_*)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_stats.txt')))
    if not files:
        return None
    with open(files[-1]) as f:
        for ln in f:
            m = re.match(r'\s*([\d.]+)%\s+calls/step\s+([\d.]+)\s+avg\s+([\d.]+) us\s+ms/step\s+([\d.]+)\s+(.*)', ln)
            if m:
                return {"kernel": m.group(5).strip(), "share_of_step_pct": float(m.group(1)), "calls_per_step": float(m.group(2)),
                        "avg_us": float(m.group(3)), "ms_per_step": float(m.group(4)), "profile": os.path.basename(files[-1])}
    return None


GCN_LAYER_SHAPES = [  # (name, C, Cout, T) at N'=128, V=25: the distinct unit_gcn shapes of configs[1]
    ('l1', 3, 64, 300), ('l2-4', 64, 64, 300), ('l5', 64, 128, 300), ('l6-7', 128, 128, 150),
    ('l8', 128, 256, 150), ('l9-10', 256, 256, 75)]


def unit_gcn_forward_roofline(device, reps=5, Np=128, V=25):
    """unit_gcn forward alone (reference agcn.py:92-109, train-mode BN, no autograd), one entry per layer shape, timed
    with HIP events on the launch stream.  Algorithmic work per SURVEY 8(d): FLOPs = theta/phi + S + aggregate +
    project (+ down); bytes = read x once + write y once.  Two roofline times are given:
      * `roofline_ms` / `frac`: per-phase-consistent -- every phase priced at the peak of the arithmetic it actually
        runs in this build (`phases`): f16x3 split MFMA (2500/3 = 833.3 TFLOP/s fp32-equivalent; agcn_chain_mode) for
        the aggregation and the conv_d projection of the chained layers, bf16x6 (2500/6 = 416.7) for the theta/phi
        projections, exact-f32 MFMA (157.3) for the score reduction, the 1x1 `down` convolution and the un-chained
        first layer; roofline = max(bytes / 8 TB/s, sum_phase FLOPs_phase / peak_phase);
      * `roofline_f32_ms` / `frac_f32`: SURVEY 8(d)'s definition, all FLOPs at the 157.3 TFLOP/s f32 matrix rate."""
    import agcn_amd  # noqa: F401
    from agcn_amd import ops
    from agcn_amd.model.agcn import unit_gcn
    from agcn_amd.graph.ntu_rgb_d import Graph
    A = Graph().A
    mode = ops._L().agcn_gemm_mode().decode()
    split_peak = (PEAK_FP32_MFMA_TFLOPS if mode == 'f32' else
                  PEAK_BF16_MFMA_TFLOPS / {'bf16x6': 6, 'bf16x3': 3, 'bf16': 1}[mode])
    chain_mode = ops._L().agcn_chain_mode().decode()       # arithmetic of the aggregate+project chain
    chain_peak = PEAK_BF16_MFMA_TFLOPS / 3 if chain_mode == 'f16x3' else split_peak
    chain_agg_split = os.environ.get('AGCN_CHAIN_F32', '0') == '0' and mode == 'bf16x6'   # aggregation on split MFMA too
    rows = []
    for name, C, Cout, T in GCN_LAYER_SHAPES:
        torch.manual_seed(0)
        m = unit_gcn(C, Cout, A).to(device).train()
        with torch.no_grad():
            m.bn.weight.fill_(1.0)
        x = torch.randn(Np, C, T, V, device=device)
        with torch.no_grad():
            for _ in range(2):
                m(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                m(x)
            e.record()
            torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        Ci = Cout // 4
        fused_adj = ops.adjacency_fused_supported(C, Ci, T, V)
        chained = C >= 32 and mode != 'f32'
        # theta/phi of the layers the persistent adjacency kernel takes (csrc/adj_ws.hip: Ci <= 32) run on f16x3
        adj_ws = fused_adj and mode != 'f32' and chain_mode == 'f16x3' and Ci in (16, 32) and C in (64, 128) \
            and os.environ.get('AGCN_ADJ_WS', '1') != '0'
        ph = {  # phase: (FLOPs, peak TFLOP/s of the arithmetic it runs in)
            'theta_phi': (Np * 6 * 2 * C * Ci * T * V,
                          chain_peak if adj_ws else (split_peak if fused_adj else PEAK_FP32_MFMA_TFLOPS)),
            'scores': (Np * 3 * 2 * V * V * Ci * T, PEAK_FP32_MFMA_TFLOPS),
            'aggregate': (Np * 3 * 2 * C * T * V * V, chain_peak if chained and chain_agg_split else PEAK_FP32_MFMA_TFLOPS),
            'project': (Np * 3 * 2 * C * Cout * T * V, chain_peak if chained else PEAK_FP32_MFMA_TFLOPS),
            'down': (Np * 2 * C * Cout * T * V if C != Cout else 0, PEAK_FP32_MFMA_TFLOPS)}
        flops = sum(f for f, _ in ph.values())
        nbytes = 4.0 * Np * (C + Cout) * T * V
        t_hbm = nbytes / 8e12 * 1e3
        t_phase = sum(f / (pk * 1e12) for f, pk in ph.values()) * 1e3
        t_f32 = flops / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
        roof, roof32 = max(t_hbm, t_phase), max(t_hbm, t_f32)
        rows.append({"layer": name, "shape": [Np, C, Cout, T, V], "ms": round(ms, 4),
                     "bound": "hbm" if t_hbm > t_phase else "mfma", "roofline_ms": round(roof, 4),
                     "frac": round(roof / ms, 4), "roofline_f32_ms": round(roof32, 4),
                     "frac_f32": round(roof32 / ms, 4), "hbm_ms": round(t_hbm, 4),
                     "phases": {k: f"{f / 1e9:.1f} GFLOP @ {pk:.1f}" for k, (f, pk) in ph.items() if f},
                     "GBps": round(nbytes / ms / 1e6, 1), "TFLOPs": round(flops / ms / 1e9, 2)})
    return rows


def host_cores():
    """CPU share of this process: the affinity mask, capped at the 16 cores a one-GPU box grants."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get('AGCN_CPU_THREADS', '16'))))


def cpu_baseline(batches=(1, 8), warmups=2, reps=5):
    """the CPU oracle (port of reference agcn.py, pinned to reference-generated fixtures, never the
    product path) doing the full training step (forward, mean CE, backward, clip_grad_norm_ 1.0, SGD momentum 0.9
    nesterov wd 1e-4) on (B,3,300,25,2) fp32, x ~ N(0,1) seed 0, for B in {1, 8}: 2 warm-ups, median of 5, all host
    cores of this box.  `value` is the better of the two batch sizes; both are in `sample`."""
    import numpy as np
    from oracle import agcn_oracle as orc
    from agcn_amd.graph.ntu_rgb_d import Graph
    cores = host_cores()
    torch.set_num_threads(cores)
    A = torch.from_numpy(Graph().A.astype(np.float32))
    res = {}
    for batch in batches:
        sd = orc.with_grad(orc.randomized_state(orc.model_param_shapes(60, 25), 1, stress=1.0))
        params = [v for k, v in sd.items() if not orc.is_buffer(k)]
        opt = torch.optim.SGD(params, lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(batch, 3, 300, 25, 2, generator=g)
        y = torch.randint(0, 60, (batch,), generator=g)

        def step():
            loss = torch.nn.functional.cross_entropy(orc.model_forward(x, sd, A, training=True), y)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
        times = []
        for i in range(warmups + reps):
            t0 = time.perf_counter()
            step()
            dt = time.perf_counter() - t0
            if i >= warmups:
                times.append(dt)
            print(f"[bench] cpu_baseline: B={batch} step {i + 1}/{warmups + reps} {dt:.2f}s", file=sys.stderr, flush=True)
        times.sort()
        res[batch] = batch / times[len(times) // 2]
    try:
        with open('/proc/cpuinfo') as f:
            model = [ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')][0]
    except (OSError, IndexError):
        model = 'unknown CPU'
    best = max(res, key=res.get)
    return {"value": round(res[best], 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "full training step (fwd, CE, bwd, clip 1.0, SGD nesterov) of the CPU oracle, fp32, median of "
                      f"{reps} after {warmups} warm-ups, " + ", ".join(f"B={b}: {v:.2f} clips/s" for b, v in res.items()) +
                      f"; {cores} threads on {model}, torch {torch.__version__}"}


def dp_rehearsal(args, device, wl, plain_ms):
    """Multi-GPU schedule on ONE GPU: a real RCCL process group of one rank, every BatchNorm stage forced onto the
    synchronised-statistics path (ops.sync_of), TrainEngine on its two-bucket overlapped gradient all-reduce.  Every
    collective is the identity, so what is measured is their count and the time they add to the step (launch + RCCL
    kernel latency + the stream synchronisation around them): the communication cost a multi-GPU step cannot hide."""
    import agcn_amd  # noqa: F401
    from agcn_amd import ops
    from agcn_amd.trainer import TrainEngine, synthetic_batch
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29544')
    # RCCL prints its version banner on stdout at the first collective: keep stdout for the one JSON line
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=device)
    orig = ops.sync_of
    ops.sync_of = lambda bn: ops.SyncBN(1, None)
    try:
        torch.manual_seed(0)
        model = build_model(args.workload)
        randomize_like_training(model, seed=0)
        model.to(device)
        eng = TrainEngine(model, base_lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, max_grad_norm=1.0, world_size=1)
        eng.rehearse = True
        eng._setup_overlap()
        data, label = synthetic_batch(args.batch, num_point=wl[2], num_class=wl[1], seed=1234, device=device)
        for _ in range(max(3, args.warmup)):
            eng.train_step(data, label)
        torch.cuda.synchronize()
        c0 = dict(ops.COLLECTIVES)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step(data, label)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        per = {k: (ops.COLLECTIVES[k] - c0[k]) / args.steps for k in c0}
    finally:
        ops.sync_of = orig
        dist.destroy_process_group()
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    return {"world": 1, "backend": "nccl (RCCL)", "collectives_per_step": {"batchnorm_statistics": per['bn'],
                                                                           "gradient_buckets": per['grad']},
            "ms_per_step_with_collectives": round(ms, 3), "ms_per_step_plain": round(plain_ms, 3),
            "exposed_ms_per_step": round(ms - plain_ms, 3),
            "note": "one rank: every collective is the identity; the difference is launch + RCCL latency + the "
                    "synchronisation around 2 collectives per unit stage and direction (a stage's statistics depend on "
                    "the previous stage's output, so they cannot be merged further) and 2 gradient buckets"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: the workload\'s, 64 for configs[1])')
    ap.add_argument('--workload', default='ntu_agcn', choices=sorted(WORKLOADS),
                    help='ntu_agcn = BASELINE configs[1]/[2] (the headline); others are the remaining configs')
    ap.add_argument('--sync-bn', choices=('auto', 'on', 'off'), default='auto',
                    help='BatchNorm statistics over the ranks.  auto (default): synchronised whenever more than one rank '
                         'runs -- the semantics of the reference\'s only multi-process path (DDP + SyncBatchNorm, '
                         'utils/processor.py:295), so the multi-GPU number pays the per-layer BN collectives the '
                         'reference pays; off: per-replica statistics (its nn.DataParallel path, :336-343)')
    ap.add_argument('--rehearse-dp', action='store_true',
                    help='one-GPU rehearsal of the multi-GPU schedule: a real RCCL group of ONE rank, every BatchNorm stage on '
                         'the synchronised path, the two-bucket gradient all-reduce; reports collectives per step and the '
                         'time they add to the step (dp_rehearsal)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()
    if args.workload.endswith('_bf16'):
        os.environ['AGCN_GEMM'] = 'bf16'          # fixed per process, read by the library on first use

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world == 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if os.environ.get('AGCN_SINGLE_GPU_RANKS'):      # test hook: all ranks on cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('AGCN_DIST_BACKEND', 'nccl')     # 'gloo' lets two ranks share one GPU in tests
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend=backend)

    import agcn_amd  # noqa: F401
    from agcn_amd.trainer import TrainEngine, synthetic_batch
    torch.manual_seed(0)                       # identical initial weights on every rank
    wl = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = wl[4]
    model = build_model(args.workload)
    randomize_like_training(model, seed=0)
    model.to(device)
    sync_bn = world > 1 and args.sync_bn != 'off'
    if sync_bn:
        from agcn_amd import dp as _dp
        model = _dp.enable_sync_bn(model, world)
    engine = TrainEngine(model, base_lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, max_grad_norm=1.0,
                         world_size=world)
    data, label = synthetic_batch(args.batch, num_point=wl[2], num_class=wl[1], seed=1234 + rank, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        engine.train_step(data, label)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = engine.train_step(data, label)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss.detach())
    if rank == 0:
        print(f"[bench] timed {args.steps} steps in {dt:.3f}s", file=sys.stderr, flush=True)

    if rank == 0:
        from agcn_amd import lib as _lib
        gemm_mode = _lib.load().agcn_gemm_mode().decode()
        dtype_label = ("f32" if gemm_mode in ('bf16x6', 'f32') else
                       "bf16" if gemm_mode == 'bf16' else "bf16x3 (reduced precision: INVALID as a headline)")
        clips = args.batch * world * args.steps
        out = {
            "metric": "skeleton-clips/sec fwd+bwd, NTU (N,3,300,25,2)" if wl[2] == 25 else
                      "skeleton-clips/sec fwd+bwd, Kinetics (N,3,300,18,2)",
            "value": round(clips / dt, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": wl[5] + ", full training step (fwd, CE, bwd, grad all-reduce, clip 1.0, "
                                   "SGD nesterov)",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "input": f"(N,3,300,{wl[2]},2)",
                       "parallelism": f"dp{world}", "bn": "sync (reference DDP semantics)" if sync_bn else "per-replica",
                       "gemm_arithmetic": gemm_mode + (" (every fp32 product = 6 bf16 MFMA products, fp32 accumulate: "
                                                       "fp32-equivalent, dropped terms < 2^-24 |ab|; the temporal "
                                                       "convolutions' and the aggregate+project chain's forward and "
                                                       "backward-data: 3 fp16 products on max-normalised operands, "
                                                       "dropped terms < 2^-22 |ab|)"
                                                       if gemm_mode == 'bf16x6' else
                                                       " (exact-f32 MFMA)" if gemm_mode == 'f32' else
                                                       " (plain bf16 MFMA operands, one product per fp32 product, fp32 "
                                                       "accumulate, fp32 storage and master weights: BASELINE "
                                                       "configs[3]; tolerance 2e-2 vs the fp32 oracle)"
                                                       if gemm_mode == 'bf16' else
                                                       " (3 bf16 products: NOT fp32-equivalent, fails the 1e-4 parity "
                                                       "bar; diagnostic mode only)")},
            "final_loss": round(final_loss, 5),
        }
        if world == 1 and not args.no_roofline and args.workload in ('ntu_agcn', 'ntu_aagcn', 'ntu_aagcn_bf16'):
            out["roofline"] = dominant_kernel_roofline(device)
            out["roofline"]["top_profiled_kernel"] = top_profiled_kernel()
            out["chain_kernel"] = chain_kernel_roofline(device)
            out["unit_gcn_fwd"] = unit_gcn_forward_roofline(device)
            if gemm_mode == 'bf16x6' and args.workload == 'ntu_agcn':
                out["exact_f32_value"] = exact_f32_value(args)     # same step, every contraction on exact-f32 MFMA
            print("[bench] roofline done", file=sys.stderr, flush=True)
        if world == 1 and (args.rehearse_dp or (not args.no_roofline and args.workload == 'ntu_agcn')):
            out["dp_rehearsal"] = dp_rehearsal(args, device, wl, dt / args.steps * 1e3)
        if world == 1 and not args.no_cpu_baseline and args.workload == 'ntu_agcn':
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
