#!/usr/bin/env python3
"""bench.py -- log_prob evaluations/second of the coupling-flow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* is one ``Flow.log_prob`` pass over one device-resident batch of synthetic standard
Gaussian rows, followed by the fp64 sum of the log-likelihood (and, for N > 1, its single
RCCL all-reduce).  The workload is BASELINE.json configs[1]: RealNVP(D=64, 8 affine coupling
layers), batch 2^20 per GPU, data-initialised weights (seed 0).  Batch rows are independent,
so ranks shard the batch with no data-path collective besides that 8-byte all-reduce
("scaling": "weak": every rank evaluates its own 2^20 rows).

One JSON line is printed by rank 0.  Besides the contract keys it carries
  roofline      the dominant libtfk kernel of the timed region: algorithmic bytes per launch
                / its mean launch duration, measured with HIP events recorded on the launch
                stream around every launch inside the timed region;
  cpu_baseline  the CPU oracle (oracle/, a C port of the reference's algorithm; the Python
                reference cannot travel to the GPU box) timed on this host's cores on a
                bounded sample of the same workload, rank 0 at N = 1 only;
  parity        max relative log_prob error of the HIP path vs the oracle on a row sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

np = torch = dist = None      # imported in main() AFTER the launcher check: the parent of an N > 1 run never
                              # touches torch, the GPU or RCCL -- it only starts the ranks and relays their output

VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4     # wave64 instructions / s (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy
FP32_PEAK_TFLOPS = 157.3       # MI355X fp32 vector peak (= fp32-input MFMA peak), same guide
PRIMING_STEPS = 40             # untimed, before the requested warm-ups (reported in the JSON)
WORKLOADS = {
    # name: (arch, D, n_layers, rows per GPU, chunk rows)
    "realnvp64": ("RealNVP", 64, 8, 1 << 20, None),          # configs[1] -- the metric's config
    "nsf64": ("CouplingRQNSF", 64, 8, 1 << 20, 1 << 18),     # configs[2]
    "realnvp256": ("RealNVP", 256, 8, 1 << 19, None),        # configs[3], one rank's shard
    "glow32": ("AffineGlow", (3, 32, 32), 3, 1 << 18, 1 << 17),  # configs[4] (3 blocks, 3.2 M params); chunks of 2^17 rows
    "lrs64": ("CouplingLRS", 64, 8, 1 << 20, 1 << 18),       # sibling preset (linear rational splines), not a BASELINE config
    "realnvp128": ("RealNVP", 128, 8, 1 << 20, None),        # between configs 2 and 4 (tuning the 128-wide kernel)
}


class KernelTimer:
    true_hidden = {}     # D -> conditioner hidden width, filled in by main() for the FLOP count

    """Wraps the libtfk entry points of torchflows_amd.native: records a HIP event pair on the
    current (= launch) stream around every call while ``active`` and keeps the algorithmic
    byte count of the launch (SURVEY.md 8(d) per-row figures x rows)."""

    def __init__(self, native):
        _heavy_imports()
        self.native = native
        self.active = False
        self.records = []   # (name, bytes, start_event, end_event)
        self.mfma_insts = {}   # kernel -> v_mfma_f32_16x16x4_f32 wave-instructions issued while active
        self.mfma_bf16_insts = {}   # kernel -> v_mfma_f32_16x16x32_bf16 wave-instructions issued while active
        for fn, byte_fn in (("affine_coupling", self._coupling_bytes(2)),
                            ("shift_coupling", self._coupling_bytes(1)),
                            ("rqs_coupling", self._rqs_bytes),
                            ("elementwise_affine", self._elementwise_bytes),
                            ("conv1x1_coupling", self._conv1x1_bytes),
                            ("permute", lambda a, k: 8 * a[0].numel()),
                            ("diag_gauss_logprob", lambda a, k: 4 * a[0].numel() + 8 * a[0].shape[0]),
                            ("flow_run", self._flow_bytes), ("flow_run_mfma", self._flow_bytes),
                            # reverse mode (csrc/tfk_bwd.hip): args (x, h, g, gld, gh, tgt, T, ...)
                            ("affine_coupling_bwd", lambda a, k: a[2].shape[0] * (a[6] * (4 + 8 + 8 + 8) + 4)),
                            ("rqs_coupling_bwd", lambda a, k: a[2].shape[0] * (a[6] * (4 + 8 + 8 * (3 * a[7] - 1)) + 4)),
                            ("shift_coupling_bwd", lambda a, k: a[0].shape[0] * a[3] * 8),
                            ("elementwise_affine_bwd",
                             lambda a, k: a[2].numel() * (12 if a[4] else 8) + (4 * a[2].shape[0] if a[4] else 0)),
                            ("diag_gauss_logprob_bwd", lambda a, k: 8 * a[0].numel() + 4 * a[0].shape[0]),
                            # fused training backward: x read, g read + written, gld
                            ("affine_coupling_train_bwd", lambda a, k: a[1].numel() * 12 + 4 * a[1].shape[0]),
                            # RQS training backward: x read, g read + written, gld, dL/dh (768) and dL/dpre (16) written
                            ("rqs_coupling_train_bwd",
                             lambda a, k: a[1].numel() * 12 + a[1].shape[0] * (4 + 4 * (784 + (16 if k.get("hid_perm") is not None else 0)))),
                            # the products that contract over the batch rows: args (A, M, B, out): A's first M columns, B read
                            ("rows_outer", lambda a, k: 4 * a[0].shape[0] * (a[1] + 16)),
                            # Glow ConvNet conditioner (csrc/tfk_convblock.hip): input read, output written
                            ("conv3x3_relu_pool_affine",
                             lambda a, k: 4 * (a[0].numel() + a[0].shape[0] * a[1].shape[0] * (a[0].shape[2] // 2) * (a[0].shape[3] // 2))),
                            ("conv1x1_frame", lambda a, k: 4 * (a[0].numel() + a[0].shape[0] * a[1].shape[0] * a[3] * a[4])),
                            ("bounded_sigmoid", lambda a, k: 8 * a[0].numel()),
                            # image flows, one launch per coupling (csrc/tfk_glow.hip): args (rows, logdet, layer, inverse)
                            ("glow_coupling", self._glow_bytes),
                            ("rows_fma", lambda a, k: 8 * a[0].numel()),
                            # flush + base density of an image program in one read-only pass (round 4)
                            ("rows_fma_gauss_logprob", lambda a, k: 4 * a[0].numel() + 8 * a[0].shape[0])):
            self._wrap(fn, byte_fn)

    @staticmethod
    def _coupling_bytes(P):
        def f(a, k):
            x, h, out = a[0], a[1], a[2]
            N, D = x.shape
            T = a[5]
            inplace = out.data_ptr() == x.data_ptr()
            row = 4 * (2 * T + T * P) + 8 if inplace else 4 * (D + T * P + D) + 8
            return N * row
        return f

    @staticmethod
    def _rqs_bytes(a, k):
        x, h, out = a[0], a[1], a[2]
        N, D = x.shape
        T, K = a[5], a[6]
        P = 3 * K - 1
        inplace = out.data_ptr() == x.data_ptr()
        return N * (4 * (2 * T + T * P) + 8 if inplace else 4 * (D + T * P + D) + 8)

    @staticmethod
    def _flow_bytes(a, k):
        # rows in (4*D) [+ rows out] [+ logdet r/w] [+ logprob w]
        x, z, logdet, logprob = a[0], a[1], a[2], a[5]
        N, D = x.shape
        acc = k.get("accumulate", False)
        return N * (4 * D + (4 * D if z is not None else 0) + ((8 if acc else 4) if logdet is not None else 0)
                    + (4 if logprob is not None else 0))

    @staticmethod
    def _conv1x1_bytes(a, k):
        x, h, out = a[0], a[1], a[2]
        N, D = x.shape
        T = a[5]
        inplace = out.data_ptr() == x.data_ptr()
        return N * ((8 * T if inplace else 8 * D) + 4 * (h.numel() // N) + 8)

    @staticmethod
    def _glow_bytes(a, k):
        from torchflows_amd import image_program
        return a[0].shape[0] * image_program.layer_cost(a[2])["bytes"]

    @staticmethod
    def _elementwise_bytes(a, k):
        x = a[0]
        N, D = x.shape
        return N * (8 * D + 8)

    def _wrap(self, name, byte_fn):
        inner = getattr(self.native, name)

        def timed(*a, **k):
            if not self.active:
                return inner(*a, **k)
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = inner(*a, **k)
            e.record()
            variant = name
            if name.endswith("_coupling") and name not in ("conv1x1_coupling", "glow_coupling"):
                variant += "[inplace]" if a[2].data_ptr() == a[0].data_ptr() else "[out-of-place]"
            flops = self._flow_flops(a, name == "flow_run_mfma") if name.startswith("flow_run") else 0
            if name == "conv3x3_relu_pool_affine":      # 2 * 9 * c_in * c_out per output position of the convolution
                n_, ci_, hh_, ww_ = a[0].shape
                variant += f"[{ci_}->{a[1].shape[0]}@{hh_}x{ww_}]"
                flops = 18 * ci_ * a[1].shape[0] * hh_ * ww_ * n_
            if name == "rows_outer":
                variant += f"[{a[1]}x16]"
            if name == "glow_coupling":
                from torchflows_amd import image_program
                L = a[2]
                variant += f"[{'conv1x1' if L.kind else 'affine'} {L.c_in}x{L.hi}x{L.wi}]"
                flops = a[0].shape[0] * image_program.layer_cost(L)["flops"]
            if name == "affine_coupling_train_bwd":
                # conditioner re-evaluation + MLP backward + weight gradients: 3 x the forward's
                # 2*(S*H + H*T*P) per row, true (unpadded) hidden width
                n_rows, d_ = a[1].shape
                h_ = KernelTimer.true_hidden.get(d_, 16)
                flops = n_rows * 6 * h_ * (d_ // 2 + d_)
            if name == "flow_run_mfma":
                self.mfma_insts[variant] = self.mfma_insts.get(variant, 0) + self._flow_mfmas(a)
                self.mfma_bf16_insts[variant] = self.mfma_bf16_insts.get(variant, 0) + self._flow_mfmas_bf16(a)
            self.records.append((variant, byte_fn(a, k), s, e, flops))
            return r
        setattr(self.native, name, timed)

    @staticmethod
    def _flow_mfmas(a):
        """v_mfma_f32_16x16x4_f32 wave-instructions of one flow_run_mfma launch (tfk_flow_mfma.h): per wave
        of 16 rows, GEMM 1 issues EPL x HT (both planes for MADE ops) and GEMM 2 tiles x k-steps."""
        x, ops = a[0], a[6]
        N, D = x.shape
        EPL = D // 8
        if not isinstance(ops, (list, tuple)):
            ops = [tuple(ops[8 * i:8 * i + 8]) for i in range(len(ops) // 8)]
        per_wave = 0
        for op in ops:
            kind, steps2 = op[0], op[2]
            HT = 1 if steps2 <= 4 else (2 if steps2 <= 8 else 4)
            f3_lean = kind in (12, 13, 14, 15) and len(op) > 4 and int(op[4]) == 256      # bf16 x 3 operands: GEMM 1 only
            if kind in (2, 3, 12, 13):
                per_wave += EPL * HT + (0 if f3_lean else (EPL // 2) * steps2)
            elif kind in (4, 5, 14, 15):
                per_wave += EPL * HT + (0 if f3_lean else (EPL // 4) * steps2)
            elif kind in (17, 18, 23, 24) and len(op) > 4 and (int(op[4]) >> 8) == 1:
                per_wave += EPL * (2 if steps2 > 4 else 1)      # GEMM 1 only: GEMM 2 runs as bf16 MFMAs (_flow_mfmas_bf16)
            elif kind in (6, 7, 17, 18):
                per_wave += EPL + 6 * EPL * steps2
            elif kind in (8, 9):
                per_wave += 2 * EPL * HT + EPL * steps2
            elif kind == 10:
                per_wave += 2 * EPL + 12 * EPL * steps2
        return per_wave * ((N + 15) // 16)

    @staticmethod
    def _flow_mfmas_bf16(a):
        """v_mfma_f32_16x16x32_bf16 wave-instructions of one launch: lean spline ops in the bf16 x 3 operand format issue
        3 per tile and hidden tile, 6 tiles per target element (8 for linear rational splines)."""
        x, ops = a[0], a[6]
        N, D = x.shape
        EPL = D // 8
        if not isinstance(ops, (list, tuple)):
            ops = [tuple(ops[8 * i:8 * i + 8]) for i in range(len(ops) // 8)]
        per_wave = 0
        for op in ops:
            if op[0] in (17, 18, 23, 24) and len(op) > 4 and (int(op[4]) >> 8) == 1:
                per_wave += (24 if op[0] >= 23 else 18) * EPL * (2 if op[2] > 4 else 1)
            elif op[0] in (12, 13, 14, 15) and len(op) > 4 and int(op[4]) == 256:
                # lean affine / shift couplings in the bf16 x 3 format (D = 256 by default): 3 per GEMM-2 tile
                per_wave += 3 * (EPL // 2 if op[0] in (12, 13) else EPL // 4)
        return per_wave * ((N + 15) // 16)

    @staticmethod
    def _flow_flops(a, mfma=False):
        """Algorithmic conditioner FLOPs of one flow_run launch: 2*(S*H + H*T*P) per row and
        coupling op; the transform's own transcendentals are not counted.
        (The matrix-core ops record ceil(H/4) instead of H; H is recovered as 4*steps rounded
        down to the true width by the caller-provided table below.)"""
        x, ops = a[0], a[6]
        N, D = x.shape
        half = D // 2
        per_row = 0
        if not isinstance(ops, (list, tuple)):          # prepacked int32[8] records
            ops = [tuple(ops[8 * i:8 * i + 8]) for i in range(len(ops) // 8)]
        for op in ops:
            kind, H = op[0], op[2]
            if mfma and kind in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 15, 17, 18, 23, 24):
                H = KernelTimer.true_hidden.get(D, 4 * H)       # the matrix-core ops record ceil(H / 4)
            P = {2: 2, 3: 2, 4: 1, 5: 1, 6: 23, 7: 23, 12: 2, 13: 2, 14: 1, 15: 1, 17: 23, 18: 23, 23: 32, 24: 32}.get(kind)
            if P is not None:
                per_row += 2 * (half * H + H * half * P)
            elif kind in (8, 9, 10):                            # MADE ops: both planes in, every element a target
                per_row += 2 * (D * H + H * D * (2 if kind != 10 else 23))
        return N * per_row

    def summary(self):
        agg = {}
        for name, nbytes, s, e, flops in self.records:
            d = agg.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["bytes"] += nbytes
            d["flops"] += flops
        for d in agg.values():
            d["flops_per_launch"] = d["flops"] // d["launches"]
            d["avg_us"] = 1e3 * d["ms"] / d["launches"]
            d["bytes_per_launch"] = d["bytes"] // d["launches"]
            d["GBps"] = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
            d["TFLOPs"] = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        return agg


def _heavy_imports():
    """numpy / torch come in on first use: the launcher parent of an N > 1 run must not touch them (see main)."""
    global np, torch, dist
    if torch is None:
        import numpy as np
        import torch
        import torch.distributed as dist


def make_flow(arch, D, n_layers):
    """seed 0, data-initialised ActNorm (one train-mode forward on 4096 host rows), eval."""
    _heavy_imports()
    import torchflows_amd as tfa
    from torchflows_amd.bijections.finite.multiscale import AffineGlow
    ctor = AffineGlow if arch == "AffineGlow" else getattr(tfa, arch)
    torch.manual_seed(0)
    flow = tfa.Flow(ctor(D, n_layers=n_layers))
    shape = D if isinstance(D, tuple) else (D,)
    flow.train()
    with torch.no_grad():
        flow.log_prob(torch.randn(512 if isinstance(D, tuple) else 4096, *shape))
    return flow.eval()


def csrc_sha256() -> str:
    """sha256 over the kernel sources (torchflows_amd/csrc/*, include/tfk.h): ties PMC summaries to a build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "torchflows_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "tfk.h")]):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


def host_cores() -> int:
    """CPUs this process may actually use: affinity mask, capped by a cgroup-v2 CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(arch, D, n_layers, flow_host, target_seconds=12.0):
    """Time the CPU oracle (a port; OpenMP over rows, all usable host cores) on a bounded
    sample: batches of 2^16 rows of the same workload until ~target_seconds have elapsed."""
    from oracle import oracle as orc
    sd = {k: v.detach().cpu().numpy() for k, v in flow_host.state_dict().items()}
    ref = orc.preset_from_state_dict(arch, D, n_layers, sd)
    threads = host_cores()
    orc.set_num_threads(threads)
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((1 << 16, D)).astype(np.float32)
    ref.log_prob(x[:4096])                                # warm-up (page in, thread pool)
    n, t0 = 0, time.perf_counter()
    while True:
        ref.log_prob(x)
        n += x.shape[0]
        dt = time.perf_counter() - t0
        if dt >= target_seconds or n >= (1 << 26):
            break
    return {"value": n / dt, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": f"{n} rows ({n // x.shape[0]} batches of 2^16) of the same workload ({arch} D={D}, "
                      f"{n_layers} coupling layers), oracle/oracle.c, OpenMP over rows on "
                      f"{threads} threads, {dt:.1f} s"}, ref


def cpu_baseline_aten(flow_host, shape, batch_rows, target_seconds=8.0):
    """SURVEY.md 8(d) "CPU comparison in the same run": this package's own ATen (pure PyTorch) path on the host
    cores -- the same op chains the reference runs on a CPU (it matches the reference's golden outputs to ~1e-7,
    tests/test_host_cpu.py / test_host_image_cpu.py); the reference itself cannot travel to the GPU box."""
    threads = host_cores()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch_rows, *shape, generator=g)
    with torch.no_grad():
        flow_host.log_prob(x[:max(1, batch_rows // 8)])
        n, t0 = 0, time.perf_counter()
        while True:
            flow_host.log_prob(x)
            n += batch_rows
            dt = time.perf_counter() - t0
            if dt >= target_seconds:
                break
    return {"value": n / dt, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": f"{n} rows ({n // batch_rows} batches of {batch_rows}) of the same workload through this "
                      f"package's ATen composite path (torchflows_amd on CPU tensors = the reference's op chains), "
                      f"torch.set_num_threads({threads}), {dt:.1f} s"}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a CHILD process
    (python -m torch.distributed.run, one rank per GPU over RCCL), relay rank 0's JSON line on stdout and
    everything else on stderr, and exit with the child's code.  This parent never imports torch and never
    touches the GPU (replacing a GPU-initialised process by exec is forbidden on this pool; a child is not)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    printed = False
    for line in proc.stdout:
        if line.startswith("{") and not printed:
            sys.stdout.write(line)
            sys.stdout.flush()
            printed = True
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and not printed:
        rc = 1
    sys.exit(rc)


def config_leg(workload, dev, timer, steps=12, warmup=3, layerwise=False):
    """One further BASELINE.json configuration on this GPU, bounded to a few seconds: `steps` Flow.log_prob + fp64-sum
    passes over the configuration's device-resident batch, each bracketed by HIP events on the launch stream (median
    reported); the dominant libtfk kernel with the numbers its roofline fraction is computed from (algorithmic FLOPs
    or bytes per launch, mean launch time); parity of the timed output against the CPU oracle (vector flows: 512 rows)
    or the reference's own fixture (config 5)."""
    from torchflows_amd import native
    from torchflows_amd.distributed import sharded_log_likelihood
    arch, D, n_layers, rows, chunk = WORKLOADS[workload]
    flow_host = make_flow(arch, D, n_layers)
    for layer in flow_host.bijection.modules():
        seq = getattr(getattr(layer, "conditioner_transform", None), "sequential", None)
        if seq is not None and not isinstance(D, tuple):
            KernelTimer.true_hidden[D] = seq[0].out_features
    flow = make_flow(arch, D, n_layers).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234)
    shape = D if isinstance(D, tuple) else (D,)
    x = torch.randn(rows, *shape, device=dev, generator=gen)
    step_rows = chunk if isinstance(D, tuple) else rows
    timer.records = []
    with torch.no_grad():
        for _ in range(warmup):
            lp, total = sharded_log_likelihood(flow, x, chunk_rows=step_rows)
        torch.cuda.synchronize()
        before = native.calls
        timer.active = True
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record()
        for i in range(steps):
            lp, total = sharded_log_likelihood(flow, x, chunk_rows=step_rows)
            marks[i + 1].record()
        torch.cuda.synchronize()
        timer.active = False
        launches = (native.calls - before) / steps
    ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    med = ms[len(ms) // 2]
    kernels = timer.summary()
    dom_name = max(kernels, key=lambda k: kernels[k]["ms"])
    dom = kernels[dom_name]
    traffic = None                 # PMC HBM bytes per launch of the dominant kernel, if taken from THESE kernel sources
    try:
        db = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if db.get("csrc_sha256") == csrc_sha256():
            traffic = db.get(workload, {}).get(dom_name)
    except (OSError, ValueError):
        pass
    out = {"workload": f"{arch}(D={D}, n_layers={n_layers}), {rows} rows" + (f" in chunks of {step_rows}" if step_rows != rows else ""),
           "rows": rows, "steps": steps, "ms_per_step": med, "min_ms": ms[0], "max_ms": ms[-1],
           "evals_per_s": rows / (med * 1e-3), "libtfk_launches_per_step": launches,
           "log_likelihood_sum": float(total.item())}
    if dom["flops"]:
        out["roofline"] = {"bound": "mfma" if dom_name.startswith("flow_run") else "valu", "kernel": dom_name,
                           "flops_per_launch": dom["flops_per_launch"], "bytes_per_launch": dom["bytes_per_launch"],
                           "avg_us": dom["avg_us"], "launches": dom["launches"], "achieved": dom["TFLOPs"],
                           "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["TFLOPs"] / FP32_PEAK_TFLOPS,
                           "hbm_GBps": dom["GBps"], "hbm_frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic}
    else:
        out["roofline"] = {"bound": "hbm", "kernel": dom_name, "bytes_per_launch": dom["bytes_per_launch"],
                           "avg_us": dom["avg_us"], "launches": dom["launches"], "achieved": dom["GBps"],
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic}
    out["kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "GBps": round(v["GBps"], 1),
                          "TFLOPs": round(v["TFLOPs"], 2)} for k, v in kernels.items()}
    if isinstance(D, tuple):
        if D == (3, 32, 32):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from golden_util import load_glow32
            gflow, fx = load_glow32()
            gflow = gflow.to(dev)
            with torch.no_grad():
                glp = gflow.log_prob(torch.from_numpy(fx["x"]).to(dev)).cpu().numpy()
            d = np.abs(glp - fx["log_prob"])
            out["parity"] = {"log_prob_max_rel_vs_reference": float(np.max(d / np.maximum(1.0, np.abs(fx["log_prob"])))),
                             "rows_checked": int(glp.shape[0])}
    else:
        from oracle import oracle as orc
        sd = {k: v.detach().cpu().numpy() for k, v in flow_host.state_dict().items()}
        ref = orc.preset_from_state_dict(arch, D, n_layers, sd)
        orc.set_num_threads(host_cores())
        idx = torch.arange(0, rows, max(rows // 512, 1), device=dev)[:512]
        lp_ref = ref.log_prob(x[idx].cpu().numpy())
        d = np.abs(lp[idx].cpu().numpy() - lp_ref)
        out["parity"] = {"log_prob_max_rel_vs_oracle": float(np.max(d / np.maximum(1.0, np.abs(lp_ref)))),
                         "rows_checked": int(idx.numel()),
                         "pass_rate_1e-5": float(np.mean(d <= 1e-5 * np.maximum(1.0, np.abs(lp_ref))))}
    if layerwise and not isinstance(D, tuple):
        # the per-layer route of the same configuration (conditioner GEMMs on PyTorch-ROCm, h through HBM): the
        # HBM-bound transform kernels north_star's 40 % target is about
        os.environ["TORCHFLOWS_AMD_FUSED"] = "0"
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        timer.records = []
        with torch.no_grad():
            sharded_log_likelihood(flow, x, chunk_rows=chunk or rows)
            torch.cuda.synchronize()
            timer.active = True
            sharded_log_likelihood(flow, x, chunk_rows=chunk or rows)
            torch.cuda.synchronize()
            timer.active = False
        os.environ["TORCHFLOWS_AMD_FUSED"] = "1"
        flow.bijection.__dict__.pop("_tfk_compiled", None)
        out["layerwise_kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                                        "bytes_per_launch": v["bytes_per_launch"], "GBps": round(v["GBps"], 1),
                                        "hbm_frac": round(v["GBps"] / HBM_PEAK_GBS, 3)}
                                    for k, v in timer.summary().items()}
    del x, flow
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="realnvp64", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (default: the config's)")
    ap.add_argument("--total-rows", type=int, default=None,
                    help="strong scaling: this many rows in total, split evenly over the ranks "
                         "(config 4: --workload realnvp256 --total-rows 4194304 --gpus 8)")
    ap.add_argument("--priming", type=int, default=PRIMING_STEPS,
                    help="untimed steps before the requested warm-ups (clock ramp after idle; reported in the JSON)")
    ap.add_argument("--stats-steps", type=int, default=100,
                    help="extra steps, timed one by one with HIP events AFTER the K timed steps (median / min / max)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true",
                    help="layer-by-layer kernels + PyTorch-ROCm conditioner GEMMs (the split path)")
    ap.add_argument("--no-sample", action="store_true", help="skip the Flow.sample throughput leg")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step (fwd + bwd + AdamW) leg")
    ap.add_argument("--chunk-rows", type=int, default=None,
                    help="rows per Flow.log_prob call of a step (default: the configuration's; 0 = the whole batch)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the bounded legs over the other BASELINE configurations (nsf64, realnvp256, glow32)")
    ap.add_argument("--no-mfma", action="store_true",
                    help="fused flow programs on the VALU interpreter (k_flow_run) instead of k_flow_run_mfma")
    args = ap.parse_args()

    if args.no_fused:
        os.environ["TORCHFLOWS_AMD_FUSED"] = "0"
    if args.no_mfma:
        os.environ["TORCHFLOWS_AMD_MFMA"] = "0"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])          # does not return
    _heavy_imports()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    # (rehearsal on a one-GPU box: TORCHFLOWS_AMD_DIST_BACKEND=gloo runs the ranks on the same card; RCCL
    # refuses two ranks per device)
    backend = os.environ.get("TORCHFLOWS_AMD_DIST_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from torchflows_amd import native
    from torchflows_amd.distributed import sharded_log_likelihood, sharded_log_likelihood_async
    native.lib()
    timer = KernelTimer(native)

    arch, D, n_layers, rows, chunk = WORKLOADS[args.workload]
    if args.chunk_rows is not None:
        chunk = args.chunk_rows or None
    rows = args.rows or rows
    if args.total_rows:
        rows = args.total_rows // world
    flow_host = make_flow(arch, D, n_layers)
    for layer in flow_host.bijection.modules():
        seq = getattr(getattr(layer, "conditioner_transform", None), "sequential", None)
        if seq is not None and not isinstance(D, tuple):
            KernelTimer.true_hidden[D] = seq[0].out_features
    flow = make_flow(arch, D, n_layers).to(dev)
    shape = D if isinstance(D, tuple) else (D,)
    DATA_BLOCK = 4096
    if args.total_rows and rows % DATA_BLOCK == 0:
        # strong scaling: ONE global data set whatever the rank count -- block b of 4096 rows is drawn from seed
        # 1234 + b, rank r holds the blocks of its row range -- so that the all-reduced log-likelihood of N ranks can be
        # held against the 1-rank value (tests/test_gpu_bench_contract.py)
        first = rank * rows // DATA_BLOCK
        x = torch.empty(rows, *shape, device=dev)
        for b in range(rows // DATA_BLOCK):
            gen = torch.Generator(device=dev).manual_seed(1234 + first + b)
            x[b * DATA_BLOCK:(b + 1) * DATA_BLOCK] = torch.randn(DATA_BLOCK, *shape, device=dev, generator=gen)
    else:
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        x = torch.randn(rows, *shape, device=dev, generator=gen)
    # the layer-by-layer path materialises h (2.9 KB per row per RQS layer): evaluate in chunks;
    # the fused programs never hold h, so they take the whole batch at once
    step_rows = (chunk or rows) if (args.no_fused or isinstance(D, tuple)) else rows

    pending = []

    def step():
        # the 8-byte all-reduce of each step stays in flight on RCCL's stream; every step's sum is
        # waited for before the clock stops (drain), so all K steps complete inside the timed region
        lp, total, work = sharded_log_likelihood_async(flow, x, chunk_rows=step_rows)
        pending.append(work)
        return lp, total

    def drain():
        for work in pending:
            work.wait()
        pending.clear()

    with torch.no_grad():
        step()                      # module load, LDS attributes, RCCL communicator: never inside the timed region,
        drain()                     # whatever --warmup says
        n_prime = args.priming if not isinstance(D, tuple) else 0      # (a config-5 step is 0.25 s: no ramp to hide)
        for _ in range(n_prime):        # ... and the clocks: the first ~20 launches after idle run 7 % slower (r02 step_stats)
            step()
        drain()
        for _ in range(args.warmup):
            step()
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timer.active = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lp, total = step()
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        timer.active = False
        # distribution of the step time: stats_steps MORE steps (not part of `value`), each bracketed by its own
        # HIP events on the launch stream; the 8-byte all-reduces stay in flight as above
        step_ms = []
        if args.stats_steps > 0:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.stats_steps + 1)]
            marks[0].record()
            for i in range(args.stats_steps):
                step()
                marks[i + 1].record()
            drain()
            torch.cuda.synchronize()
            step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.stats_steps))
    rows_total = world * rows
    rank_ms, rank_sums, rank_devices = [1e3 * elapsed / args.steps], None, [local_dev]
    if world > 1:
        # every rank's own step time (a straggler shows as max >> min) and its shard of the log-likelihood (the shards
        # must add up to the all-reduced total): two small all-gathers AFTER the timed region
        mine = torch.tensor([elapsed, float(native.sum_f32(lp).item()), float(torch.cuda.current_device())],
                            dtype=torch.float64, device=dev)
        both = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(both, mine)
        rank_ms = [1e3 * float(b[0]) / args.steps for b in both]
        rank_sums = [float(b[1]) for b in both]
        rank_devices = [int(b[2]) for b in both]             # the HIP device every rank is bound to (rank r -> device r)
        t = torch.tensor([elapsed, float(rows)], dtype=torch.float64, device=dev)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        r = t[1:].clone()
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        elapsed, rows_total = float(t[0].item()), int(r.item())

    if rank == 0:
        kernels = timer.summary()
        dom_name = max(kernels, key=lambda k: kernels[k]["ms"])
        dom = kernels[dom_name]
        # PMC counters need their own rocprofv3 passes: the committed summary is read -- but only if it was taken
        # from THESE kernel sources (profiles/traffic.json carries the sha256 of torchflows_amd/csrc/ at the time of
        # the PMC run, tools/summarize_profile.py); a kernel change leaves traffic / valu_insts null, not stale
        traffic_db, traffic_stale = {}, None
        try:
            db = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic_stale = db.get("csrc_sha256") != csrc_sha256()
            if not traffic_stale and rows == WORKLOADS[args.workload][3]:
                traffic_db = db.get(args.workload, {})
        except (OSError, ValueError):
            pass
        traffic = traffic_db.get(dom_name)
        if dom_name.startswith("flow_run"):
            # conditioner fused in-kernel: h never reaches HBM.  SURVEY.md 8(d): report the fused
            # kernel against the fp32 matrix / vector peak (the same 157.3 TFLOP/s) and say so.
            # The GEMMs of flow_run_mfma run on the matrix cores ("mfma"); what actually limits
            # both kernels is vector-ALU issue of the transform's exp / log / divide ("limiter").
            valu = traffic_db.get(dom_name + ":valu_insts")
            mfma_pmc = traffic_db.get(dom_name + ":mfma_insts")
            roofline = {"bound": "mfma" if dom_name == "flow_run_mfma" else "valu",
                        "achieved": dom["TFLOPs"], "peak": FP32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": dom["TFLOPs"] / FP32_PEAK_TFLOPS, "traffic": traffic,
                        "limiter": "fp32-datapath",
                        "note": "fused flow program (conditioner in-kernel" +
                                (", GEMMs as v_mfma_f32_16x16x4_f32" if dom_name == "flow_run_mfma" else "") +
                                "); achieved = algorithmic conditioner FLOPs 2*(S*H+H*T*P) per row-layer "
                                "/ launch time (the transform's transcendentals are not counted); peak = "
                                "dense fp32 matrix peak; the kernel is limited by the SIMD FP32 datapath that "
                                "vector instructions and f32-input MFMAs share, see fp32_datapath_frac",
                        "hbm_GBps": dom["GBps"], "hbm_frac": dom["GBps"] / HBM_PEAK_GBS}
            if valu is not None:
                # wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, profiles/) / live duration
                # against 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles at 2.4 GHz
                rate = valu / (dom["avg_us"] * 1e-6)
                roofline["valu_insts_per_launch"] = valu
                roofline["valu_issue_frac"] = rate / VALU_ISSUE_PEAK
                n_mfma = timer.mfma_insts.get(dom_name, 0) / max(dom["launches"], 1)
                n_bf16 = timer.mfma_bf16_insts.get(dom_name, 0) / max(dom["launches"], 1)
                roofline["mfma_f32_insts_per_launch_model"] = n_mfma
                if n_bf16:
                    roofline["mfma_bf16_insts_per_launch_model"] = n_bf16
                if mfma_pmc is not None:
                    # measured: rocprofv3 SQ_INSTS_VALU_MFMA_F32 / _BF16 (tools/profile.sh pass 5); the analytic counts
                    # must agree
                    roofline["mfma_insts_per_launch_pmc"] = mfma_pmc
                    roofline["mfma_insts_model_vs_pmc"] = n_mfma / mfma_pmc if mfma_pmc else None
                    bf16_pmc = traffic_db.get(dom_name + ":mfma_bf16_insts")
                    if bf16_pmc:
                        roofline["mfma_bf16_insts_per_launch_pmc"] = bf16_pmc
                        roofline["mfma_bf16_insts_model_vs_pmc"] = n_bf16 / bf16_pmc
                gui = traffic_db.get(dom_name + ":grbm_gui_active")
                busy_mfma = traffic_db.get(dom_name + ":mfma_busy_cycles")
                if gui and busy_mfma is not None:
                    # matrix-pipe utilisation as the profiler's MfmaUtil defines it: SQ_VALU_MFMA_BUSY_CYCLES over
                    # (kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs; COEXEC = cycles in which a vector
                    # instruction issued while the matrix pipe was busy (0 for the f32-input MFMA: it occupies the
                    # SIMD's FP32 datapath, nothing runs beside it)
                    simd_cycles = gui / 8.0 * 256 * 4
                    roofline["mfma_util"] = busy_mfma / simd_cycles
                    roofline["mfma_coexec_frac"] = (traffic_db.get(dom_name + ":mfma_coexec_cycles") or 0.0) / simd_cycles
                    trans = traffic_db.get(dom_name + ":trans_insts") or 0.0
                    vec = valu - (mfma_pmc or 0.0) - (traffic_db.get(dom_name + ":mfma_bf16_insts") or 0.0)   # SQ_INSTS_VALU counts the MFMAs too
                    roofline["vector_insts_per_launch"] = vec
                    # SIMD cycles accounted for: 4 per vector instruction (+4 for a transcendental) + the matrix pipe, less
                    # the cycles the counter saw both at work (bf16 MFMAs co-issue; f32-input ones never do)
                    coexec = traffic_db.get(dom_name + ":mfma_coexec_cycles") or 0.0
                    roofline["fp32_datapath_frac"] = (4.0 * vec + 4.0 * trans + busy_mfma - coexec) / simd_cycles
                elif n_mfma:
                    busy = 4.0 * (valu - n_mfma) + 32.0 * n_mfma
                    roofline["mfma_insts_per_launch"] = n_mfma
                    roofline["fp32_datapath_frac"] = busy / (dom["avg_us"] * 1e-6 * 256 * 4 * 2.4e9)
        elif dom_name.startswith("glow_coupling"):
            # a whole convolutional coupling in one launch: the conv blocks are direct convolutions on the vector ALUs
            # (v_pk_fma_f32: c_out = 8 / 4 would waste half / three quarters of a 16-wide MFMA tile), so the launch is
            # priced against the fp32 vector peak; FLOPs = what the kernel executes (conv blocks on the windows the
            # source image reaches, Linear at K = 16), fewer than the reference's formulation of the same values
            roofline = {"bound": "valu", "achieved": dom["TFLOPs"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": dom["TFLOPs"] / FP32_PEAK_TFLOPS, "traffic": traffic,
                        "flops_per_launch": dom["flops"] // dom["launches"],
                        "note": "tfk_glow_coupling: ConvNet conditioner (LDS-resident) + Linear (MFMA) + bounded "
                                "sigmoid + transform in one launch; achieved = executed multiply-adds x 2 / launch "
                                "time against the fp32 vector peak (packed v_pk_fma_f32; tools/micro/rates.hip measures "
                                "111 TFLOP/s for a pure stream of them)",
                        "hbm_GBps": dom["GBps"], "hbm_frac": dom["GBps"] / HBM_PEAK_GBS}
        elif dom_name.startswith("conv3x3_relu_pool_affine"):
            # direct convolution on the vector ALUs (c_out = 8 would waste half of a 16-wide MFMA tile and
            # the fp32 MFMA peak equals the vector peak): priced against the fp32 vector peak
            roofline = {"bound": "valu", "achieved": dom["TFLOPs"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": dom["TFLOPs"] / FP32_PEAK_TFLOPS, "traffic": traffic,
                        "note": "conv3x3 + ReLU + MaxPool + BatchNorm-affine block of the ConvNet conditioner, one "
                                "launch: 2*9*c_in*c_out FLOPs per convolution output / launch time against the "
                                "fp32 vector peak",
                        "hbm_GBps": dom["GBps"], "hbm_frac": dom["GBps"] / HBM_PEAK_GBS}
        else:
            roofline = {"bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic}
        roofline.update({"kernel": dom_name, "bytes_per_launch": dom["bytes_per_launch"],
                         "avg_us": dom["avg_us"], "launches": dom["launches"]})
        result = {
            "metric": "log_prob evals/sec (RealNVP D=64)" if args.workload == "realnvp64"
                      else f"log_prob evals/sec ({args.workload})",
            "value": rows_total * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.total_rows else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{arch}(D={D}, n_layers={n_layers}) Flow.log_prob + fp64 sum"
                                   f"{' + 8-byte RCCL all-reduce' if world > 1 else ''}, "
                                   f"{rows} standard-Gaussian rows per GPU resident in HBM, "
                                   f"data-initialised weights (seed 0)",
                       "rows_per_gpu": rows, "rows_total": rows_total,
                       "parallelism": f"batch-sharded replicas x{world}"},
            "priming_steps": 1 + n_prime,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "collective_backend": (backend if world > 1 else None),
            "rows_all_ranks": rows_total,
            "log_likelihood_sum": float(total.item()),
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
            "log_likelihood_shards": rank_sums,
            "rank_devices": rank_devices, "devices_visible": torch.cuda.device_count(),
            "step_stats": ({"steps": len(step_ms), "median_ms": step_ms[len(step_ms) // 2], "min_ms": step_ms[0],
                            "max_ms": step_ms[-1], "p10_ms": step_ms[len(step_ms) // 10],
                            "p90_ms": step_ms[(9 * len(step_ms)) // 10],
                            "evals_per_s_at_median": rows / (step_ms[len(step_ms) // 2] * 1e-3) * world,
                            "note": "per-step HIP-event times of stats_steps further steps on rank 0 (not part of value)"}
                           if step_ms else None),
            "traffic_source": (None if traffic_stale is None else
                               ("stale: profiles/traffic.json was taken from other kernel sources -- traffic / "
                                "valu_insts withheld" if traffic_stale else "profiles/traffic.json (csrc sha256 matches)")),
            "roofline": roofline,
            "kernels": {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                            "GBps": round(v["GBps"], 1)} for k, v in kernels.items()},
            "libtfk_ms_per_step": sum(v["ms"] for v in kernels.values()) / args.steps,
        }
        if world == 1 and dom_name.startswith("flow_run"):
            # the same workload layer by layer (one libtfk kernel per reference layer, conditioner
            # GEMMs on PyTorch-ROCm): its transform kernels are the HBM-bound ones, so their
            # roofline is reported next to the fused program's
            os.environ["TORCHFLOWS_AMD_FUSED"] = "0"
            flow.bijection.__dict__.pop("_tfk_compiled", None)
            timer.records = []
            lw_rows = chunk or rows
            with torch.no_grad():
                sharded_log_likelihood(flow, x, chunk_rows=lw_rows)
                torch.cuda.synchronize()
                timer.active = True
                t1 = time.perf_counter()
                for _ in range(3):
                    sharded_log_likelihood(flow, x, chunk_rows=lw_rows)
                torch.cuda.synchronize()
                lw_elapsed = time.perf_counter() - t1
                timer.active = False
            os.environ["TORCHFLOWS_AMD_FUSED"] = "1"
            lw = timer.summary()
            lw_name = max(lw, key=lambda k: lw[k]["ms"])
            tr = traffic_db.get(lw_name)
            result["roofline_layerwise"] = {
                "bound": "hbm", "achieved": lw[lw_name]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": lw[lw_name]["GBps"] / HBM_PEAK_GBS, "traffic": tr, "kernel": lw_name,
                "bytes_per_launch": lw[lw_name]["bytes_per_launch"], "avg_us": lw[lw_name]["avg_us"],
                "launches": lw[lw_name]["launches"], "evals_per_s": rows * 3 / lw_elapsed,
                "kernels": {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                                "bytes_per_launch": v["bytes_per_launch"], "GBps": round(v["GBps"], 1),
                                "hbm_frac": round(v["GBps"] / HBM_PEAK_GBS, 3)} for k, v in lw.items()},
                "note": "same workload with TORCHFLOWS_AMD_FUSED=0 (bench.py --no-fused), 3 steps; `kernels` lists every "
                        "per-layer kernel's HBM fraction (affine_coupling[inplace] is the coupling transform pass)"}
        if world == 1 and not args.no_sample:
            # sample throughput next to log_prob (Flow.sample = base draw +
            # bijection.inverse with log-det, flows.py:117-146), same rows, not part of `value`
            srows = step_rows
            with torch.no_grad():
                flow.sample((srows,), return_log_prob=True)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(3):
                    xs, lps = flow.sample((srows,), return_log_prob=True)
                torch.cuda.synchronize()
                s_elapsed = time.perf_counter() - t2
            result["sample"] = {"value": srows * 3 / s_elapsed, "unit": "samples/s", "rows": srows,
                                "note": "Flow.sample((rows,), return_log_prob=True): torch.randn base "
                                        "draw + inverse flow program, 3 calls"}
            del xs, lps
        if world == 1 and not args.no_train and not isinstance(D, tuple):
            # one maximum-likelihood step on the HIP path = Flow.log_prob with
            # autograd (layer kernels, out of place) + reverse-mode kernels + conditioner GEMM
            # backward on PyTorch-ROCm + AdamW.  Not part of `value`.
            trows = min(rows, chunk or rows, 1 << 18)
            xt = x[:trows]
            flow.train()
            from torchflows_amd.utils import make_adamw
            opt = make_adamw(flow.parameters(), 1e-4)

            wt = torch.ones(trows, device=xt.device)

            def train_step():                    # the step Flow.fit runs (flows.py: _base_batch_loss, reference :199-224)
                opt.zero_grad(set_to_none=True)
                loss = flow._base_batch_loss((xt, wt), reduction=torch.mean, use_regularization=True)
                loss.backward()
                opt.step()
                return loss

            for _ in range(3):                   # lazy state: optimiser moments, index maps, allocator blocks
                train_step()
            torch.cuda.synchronize()
            n_train_steps = 20
            t3 = time.perf_counter()
            for _ in range(n_train_steps):
                loss = train_step()
            torch.cuda.synchronize()
            t_elapsed = time.perf_counter() - t3
            timer.records = []
            timer.active = True                  # per-kernel HIP events: two more steps, not timed above
            for _ in range(2):
                train_step()
            torch.cuda.synchronize()
            timer.active = False
            flow.eval()
            tk = timer.summary()
            bwd = {k: v for k, v in tk.items() if k.endswith("_bwd") and not k.endswith("_train_bwd")}
            top = max(bwd, key=lambda k: bwd[k]["ms"]) if bwd else None
            result["train"] = {
                "value": trows * n_train_steps / t_elapsed, "unit": "samples/s", "rows": trows, "steps": n_train_steps,
                "ms_per_step": 1e3 * t_elapsed / n_train_steps, "loss": float(loss.detach()),
                "optimizer": type(opt).__name__,
                "libtfk_ms_per_step": sum(v["ms"] for v in tk.values()) / 2,
                "kernels": {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                                "GBps": round(v["GBps"], 1)} for k, v in tk.items()},
                "note": "the step Flow.fit runs, eager (no hipGraph): _base_batch_loss with autograd + backward + "
                        "AdamW.  libtfk layer kernels, single-op MFMA flow programs for the coupling forward, fused "
                        "coupling backward (conditioner, transform, MLP and weight-gradient sums in one launch) where "
                        "supported, else reverse-mode kernels + PyTorch-ROCm GEMMs; the parameters live in one buffer "
                        "(flat_optim.py): operands by one gather, the L2 penalty inside the chain's autograd node, "
                        "gradients as slices of one buffer, AdamW as torch's own _foreach calls on that buffer; "
                        "libtfk_ms_per_step is the share spent in libtfk kernels (HIP events over two extra steps)"}
            if args.workload == "realnvp64":
                # the same step captured once into a hipGraph and replayed (TORCHFLOWS_AMD_GRAPH=1 in
                # Flow.fit).  Measured in a child process: an invalidated capture crashes the process
                # on this stack instead of raising, and must not take the bench line with it.
                import subprocess
                try:
                    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "graph_probe.py"), "full"],
                                          capture_output=True, text=True, timeout=240)
                    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("full:")]
                    if proc.returncode == 0 and line:
                        ms = float(line[-1].split()[1])
                        result["train"]["hipgraph"] = {"ms_per_step": ms, "value": trows / (ms * 1e-3),
                                                       "unit": "samples/s", "steps": 20,
                                                       "note": "tools/graph_probe.py: one captured step "
                                                               "(fwd + bwd + AdamW) replayed 20 times"}
                    else:
                        result["train"]["hipgraph"] = {"error": f"exit code {proc.returncode}: "
                                                                + proc.stderr.strip()[-200:]}
                except (subprocess.TimeoutExpired, OSError, ValueError) as exc:
                    result["train"]["hipgraph"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
                # Flow.fit itself at the reference's default batch size (1024 rows: pure host work per step), eager against
                # the default ("auto": the fully fused step captured after two eager ones) -- also in a child process
                try:
                    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fit_probe.py"), "64"],
                                          capture_output=True, text=True, timeout=240)
                    got = {}
                    for ln in proc.stdout.splitlines():
                        if "TORCHFLOWS_AMD_GRAPH=" in ln and " ms per step" in ln:
                            mode = ln.split("TORCHFLOWS_AMD_GRAPH=")[1].split()[0]
                            got[mode] = float(ln.split(" = ")[1].split()[0])          # (the later, warm run of each mode)
                    if proc.returncode == 0 and "0" in got and "auto" in got:
                        result["train"]["fit_batch_1024"] = {
                            "eager_ms_per_step": got["0"], "default_ms_per_step": got["auto"], "steps": 192,
                            "note": "tools/fit_probe.py: Flow.fit(x[65536, 64], n_epochs=3) wall time / 192 steps, "
                                    "TORCHFLOWS_AMD_GRAPH=0 against the default"}
                    else:
                        result["train"]["fit_batch_1024"] = {"error": f"exit code {proc.returncode}: "
                                                                      + proc.stderr.strip()[-200:]}
                except (subprocess.TimeoutExpired, OSError, ValueError, IndexError) as exc:
                    result["train"]["fit_batch_1024"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
                # the reference's only published timings: the tqdm rates in its notebooks (hardware
                # not stated) -- the same calls on this build, a few seconds each, in a child process
                try:
                    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "notebook_probe.py"), "3.0"],
                                          capture_output=True, text=True, timeout=300)
                    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("NOTEBOOK_JSON ")]
                    if proc.returncode == 0 and line:
                        result["train"].update(json.loads(line[-1][len("NOTEBOOK_JSON "):]))
                    else:
                        result["train"]["notebook_error"] = f"exit code {proc.returncode}: " + proc.stderr.strip()[-300:]
                except (subprocess.TimeoutExpired, OSError, ValueError) as exc:
                    result["train"]["notebook_error"] = f"{type(exc).__name__}: {exc}"[:300]
                # config 5's model in TRAINING at Flow.fit's default batch size: the ConvNet conditioner on ATen / MIOpen
                # (the route of rounds 1-3) against csrc/tfk_convtrain.hip, eager and captured (child process, ~20 s)
                try:
                    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "glow_train_probe.py"), "20"],
                                          capture_output=True, text=True, timeout=300)
                    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("GLOW_TRAIN_JSON ")]
                    if proc.returncode == 0 and line:
                        result["train"]["glow32"] = json.loads(line[-1][len("GLOW_TRAIN_JSON "):])
                    else:
                        result["train"]["glow32"] = {"error": f"exit code {proc.returncode}: " + proc.stderr.strip()[-300:]}
                except (subprocess.TimeoutExpired, OSError, ValueError) as exc:
                    result["train"]["glow32"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            rb = tk.get("rqs_coupling_train_bwd")
            if rb is not None:
                result["train"]["roofline_fused_bwd"] = {
                    "bound": "hbm", "kernel": "rqs_coupling_train_bwd", "achieved": rb["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": rb["GBps"] / HBM_PEAK_GBS, "avg_us": rb["avg_us"],
                    "bytes_per_launch": rb["bytes_per_launch"],
                    "note": "conditioner re-evaluated in the kernel; dL/dh (3 KB per row) written once; the kernel "
                            "is bound by vector-ALU issue of the spline backward, not by HBM"}
            fb = tk.get("affine_coupling_train_bwd")
            if fb is not None:
                result["train"]["roofline_fused_bwd"] = {
                    "bound": "mfma", "kernel": "affine_coupling_train_bwd", "achieved": fb["TFLOPs"],
                    "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fb["TFLOPs"] / FP32_PEAK_TFLOPS,
                    "avg_us": fb["avg_us"], "hbm_GBps": fb["GBps"], "hbm_frac": fb["GBps"] / HBM_PEAK_GBS,
                    "note": "algorithmic FLOPs 6*H*(S + T*P) per row (conditioner re-evaluation, MLP backward, "
                            "weight gradients; v_mfma_f32_16x16x4_f32), launch time includes the column-sum "
                            "kernel that follows"}
            if top is not None:
                result["train"]["roofline_bwd"] = {
                    "bound": "hbm", "kernel": top, "achieved": bwd[top]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": bwd[top]["GBps"] / HBM_PEAK_GBS,
                    "bytes_per_launch": bwd[top]["bytes_per_launch"], "avg_us": bwd[top]["avg_us"]}
        if world == 1 and not args.no_cpu_baseline and not isinstance(D, tuple):
            base, ref = cpu_baseline(arch, D, n_layers, flow_host)
            result["cpu_baseline"] = base
            result["cpu_baseline_aten"] = cpu_baseline_aten(flow_host, (D,), 1 << 14)
            idx = torch.arange(0, rows, max(rows // 2048, 1), device=dev)[:2048]
            lp_ref = ref.log_prob(x[idx].cpu().numpy())
            d = np.abs(lp[idx].cpu().numpy() - lp_ref)
            err = np.max(d / np.maximum(1.0, np.abs(lp_ref)))
            result["parity"] = {"log_prob_max_rel_vs_oracle": float(err), "rows_checked": int(idx.numel()),
                                "pass_rate_1e-5": float(np.mean(d <= 1e-5 * np.maximum(1.0, np.abs(lp_ref))))}
        if world == 1 and not args.no_cpu_baseline and isinstance(D, tuple):
            # config 5: the oracle's C port covers the image LAYERS (masks, squeeze, 1x1 convolution), not the ConvNet
            # conditioner, so the CPU leg is this package's ATen path on the host cores and parity is taken against
            # the REFERENCE's own outputs for this model (tests/golden/flow_glow_3x32x32.npz, make_golden.py gen_glow32)
            result["cpu_baseline"] = cpu_baseline_aten(flow_host, D, 256, target_seconds=12.0)
            result["cpu_baseline_aten"] = result["cpu_baseline"]
            if D == (3, 32, 32):
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from golden_util import load_glow32
                gflow, fx = load_glow32()
                gflow = gflow.to(dev)
                with torch.no_grad():
                    glp = gflow.log_prob(torch.from_numpy(fx["x"]).to(dev)).cpu().numpy()
                d = np.abs(glp - fx["log_prob"])
                result["parity"] = {"log_prob_max_rel_vs_reference": float(np.max(d / np.maximum(1.0, np.abs(fx["log_prob"])))),
                                    "rows_checked": int(glp.shape[0]),
                                    "pass_rate_1e-5": float(np.mean(d <= 1e-5 * np.maximum(1.0, np.abs(fx["log_prob"])))),
                                    "note": "HIP path vs the reference's own log_prob of the config-5 model "
                                            "(seed-0 weights pinned by sha256, data-dependent state from the fixture)"}
        if world == 1 and args.workload == "realnvp64" and not args.no_configs and not args.no_fused:
            result["configs"] = {}
            for name in ("nsf64", "realnvp256", "glow32"):
                t_leg = time.perf_counter()
                try:
                    leg = config_leg(name, dev, timer, layerwise=(name == "nsf64"))
                except Exception as exc:                      # a leg must not take the headline line with it
                    leg = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                leg["leg_seconds"] = time.perf_counter() - t_leg
                result["configs"][name] = leg
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
