#!/usr/bin/env python3
"""bench.py -- clouds/sec, fwd+bwd, pointnet_pp_vonMises, N=1024 (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks: the parent -- before
it has made a single GPU call -- runs `python -m torch.distributed.run ... bench.py <same arguments>` as a child process
and exits with its code (never an exec of a process that has touched the GPU).  Under torchrun it is a plain rank.

One step = zero_grad + forward (device-side centre sampling, kNN grouping, fused MLP, head) + single-peak
von-Mises KL + backward + [one flat-gradient all-reduce when N > 1] + fused Adam, on a batch of B=32 synthetic
clouds per GPU that is already resident in HBM.  float32 end to end (the large products as six exact bf16 x bf16 partial products of
three-way operand splits, float32 accumulate: --f32-products; `f32_mfma_variant` = the same step on v_mfma_f32_32x32x2_f32), weak scaling.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      -- the kernel that took the most time: achieved rate from HIP events recorded by the library
                   around every launch in a SEPARATE instrumented pass (never while throughput is timed)
  cpu_baseline  -- the CPU oracle (float32 restatement of the reference step) timed on the host cores on a
                   bounded sample of the same workload (rank 0, N=1 only); baseline only, not a target.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

N_POINTS = 1024
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: BF16 dense (the opt-in bf16-operand variant's kernels)
# float32 products formed as six exact bf16 x bf16 partial products (csrc/gemm_wsf3_kernels.hip): float32-equivalent FLOP/s of the bf16 pipe
MFMA_SPLIT_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
SPLIT_KERNELS = ("gemm_wsf3_kernel", "gemm_wsd3_kernel", "gemm_mid3_kernel", "gemm_wsf03_kernel")
FLOPS_PER_CLOUD = 0.8542e9     # 3 x 2 x 142,369,280 MAC, forward + backward, independent of N
BYTES_PER_CLOUD = 34.6e6       # 5 E + 3 G float32 words + xyz / indices


def is_capture_error(e: BaseException) -> bool:
    """True for what a failed stream capture raises: HIP's capture status codes (hipErrorStreamCaptureUnsupported /
    ...Invalidated / ...Merge / ...Unmatched / ...Unjoined / ...Isolation / ...Implicit / ...WrongThread, hipErrorCapturedEvent --
    "operation not permitted when stream is capturing", "operation failed due to a previous error during capture", ...) and
    torch's own capture checks ("... must be captured on a non-default stream"), all of which say capture / capturing / captured.
    Anything else -- TypeError, ValueError, a PNPP status, a HIP fault, an NCCL error, autograd's "backward through the graph a
    second time" -- is a bug or a fault and propagates."""
    if not isinstance(e, RuntimeError) or isinstance(e, NotImplementedError):
        return False
    return "captur" in str(e).lower()


# Data-parallel schedules.  The host-issued ones run on any backend and are rehearsed on two ranks by tests/: they are built, timed
# and MEASURED first.  The captured ones (RCCL's launches as nodes of the step's hipGraph) have only ever executed with a one-rank
# communicator (the builder's box has one GPU), so they are opt-in (PNPP_DP_CAPTURED=1) and, when opted in, tried only AFTER the
# measurement of the best host-issued schedule is in hand, each under a bounded wait (ScheduleTrial).
SAFE_SCHEDULES = ("single", "overlap")
CAPTURED_SCHEDULES = ("captured_single", "captured_overlap")
DP_SCHEDULES = CAPTURED_SCHEDULES + SAFE_SCHEDULES
SCHEDULE_MODE_PREFIX = {"captured_overlap": "ONE hipGraph (fwd + bwd, the two", "captured_single": "ONE hipGraph (fwd + bwd + the captured",
                        "overlap": "two hipGraphs", "single": "hipGraph(fwd+loss+bwd) + eager", "eager": "eager"}


def run_bounded(fn, limit_s, thread_setup=None):
    """Runs fn() on a helper thread and waits at most limit_s for it: ("ok", value) | ("error", exception) | ("timeout", None).
    A timed-out helper is abandoned (daemon thread): whatever it waits for -- a device that no longer answers, a collective whose
    peers never arrive -- must not be touched again by the caller."""
    import threading
    box = {}

    def target():
        try:
            if thread_setup is not None:
                thread_setup()
            box["value"] = fn()
        except BaseException as e:   # noqa: BLE001 -- handed to the caller, which decides
            box["error"] = e

    th = threading.Thread(target=target, daemon=True)
    th.start()
    th.join(limit_s)
    if th.is_alive():
        return "timeout", None
    if "error" in box:
        return "error", box["error"]
    return "ok", box.get("value")


class Ctrl:
    """Control plane between the ranks: agreement on small CPU values over gloo, independent of the GPU streams and of the RCCL
    communicator -- a rank whose device hangs can still tell the others, and nobody waits on the communicator that hung."""

    def __init__(self, world):
        self.world, self.group, self.device = world, None, None
        if world > 1:
            import datetime
            import torch.distributed as tdist
            # always a group of its own (also when the data plane is gloo): a helper thread may be stuck inside a data-plane collective
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: never resolve the host name (it may not resolve here)
            try:
                self.group = tdist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
            except Exception as e:   # no gloo transport on this box: agree over the data plane (enough for the host-issued schedules)
                print(f"[bench] control plane falls back to the {tdist.get_backend()} group: {type(e).__name__}: {e}", file=sys.stderr)
                self.group = None
                if tdist.get_backend() == "nccl":
                    self.device = torch.device("cuda", torch.cuda.current_device())

    def _reduce(self, x, op):
        if self.world == 1:
            return float(x)
        import torch.distributed as tdist
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        tdist.all_reduce(t, op=getattr(tdist.ReduceOp, op), group=self.group)
        return float(t)

    def all_ok(self, flag) -> bool:
        return self._reduce(1.0 if flag else 0.0, "MIN") >= 1.0

    def max(self, x) -> float:
        return self._reduce(x, "MAX")


class ScheduleTrial:
    """Builds and times data-parallel schedules so that the run cannot lose its measurement.

    build(name) -> (step, mode, local)   may raise; time_steps(step, n) -> seconds on this rank for n steps (fenced).
    safe(names):     built and timed in the open.  An exception propagates (a bug or a fault in the step must exit non-zero with its
                     message, never hide behind a slower schedule); a schedule whose capture fell back to another form is recorded.
    optional(names): each build and each timing under run_bounded.  Any failure on any rank (error, timeout, wrong form) is agreed
                     over the control plane and the schedule is abandoned by every rank; after a TIMEOUT -- or an error that is not a
                     capture error -- the device or the communicator may be unusable: `poisoned` is set, nothing further is tried,
                     and the caller must finish on what it has already measured without touching the GPU again."""

    def __init__(self, build, time_steps, ctrl, bounded_s=60.0, thread_setup=None, log=None):
        self.build, self.time_steps, self.ctrl, self.bounded_s, self.thread_setup = build, time_steps, ctrl, bounded_s, thread_setup
        self.log = log or (lambda msg: print(f"[bench] {msg}", file=sys.stderr))
        self.built, self.trial_ms, self.failed, self.poisoned = {}, {}, {}, False

    def _form_ok(self, name, mode):
        return mode.startswith(SCHEDULE_MODE_PREFIX.get(name, ""))

    def safe(self, names, n_warm=5, n=30):
        for name in names:
            cand = self.build(name)
            if not self.ctrl.all_ok(self._form_ok(name, cand[1])):
                self.failed[name] = "capture fell back: " + cand[1][:60]
                continue
            for _ in range(n_warm):
                cand[0]()
            self.built[name] = cand
            self.trial_ms[name] = self.ctrl.max(self.time_steps(cand[0], n)) / n * 1e3
        return self

    def optional(self, names, n_warm=5, n=30):
        for name in names:
            if self.poisoned:
                self.failed.setdefault(name, "not tried: an earlier optional schedule left the device or the communicator unusable")
                continue
            status, val = run_bounded(lambda: self.build(name), self.bounded_s, self.thread_setup)
            mine = status == "ok" and self._form_ok(name, val[1])
            hard = status == "timeout" or (status == "error" and not is_capture_error(val))
            why = (f"timed out after {self.bounded_s:.0f} s while building" if status == "timeout"
                   else f"{type(val).__name__}: {str(val)[:160]}" if status == "error"
                   else None if mine else "capture fell back: " + val[1][:60])
            if self.ctrl.all_ok(mine):
                def warm_and_time(step=val[0]):
                    for _ in range(n_warm):
                        step()
                    return self.time_steps(step, n)
                status, t = run_bounded(warm_and_time, self.bounded_s, self.thread_setup)
                mine = status == "ok"
                hard = hard or not mine
                why = (f"timed out after {self.bounded_s:.0f} s in its first replays" if status == "timeout"
                       else f"{type(t).__name__}: {str(t)[:160]}" if status == "error" else None)
                if self.ctrl.all_ok(mine):
                    self.built[name] = val
                    self.trial_ms[name] = self.ctrl.max(t) / n * 1e3
                    continue
            self.failed[name] = why or "another rank could not build or run it"
            self.log(f"{name}: abandoned by every rank ({self.failed[name]})")
            # a failure on ANOTHER rank leaves this rank's helper waiting for peers that will not come: that is a hang too
            if not self.ctrl.all_ok(not hard):
                self.poisoned = True
        return self

    def best(self, among=None):
        pool = {k: v for k, v in self.trial_ms.items() if among is None or k in among}
        return min(pool, key=pool.get) if pool else None

    def report(self, chosen):
        return {"candidates_ms": dict(self.trial_ms), "chosen": chosen, **({"not_built": dict(self.failed)} if self.failed else {})}


def build_step(model, opt, xyz, mu_gt, kappa_gt, world, use_graph, collective=True, schedule=None):
    """Returns (step, launch_mode, step_without_collective).  zero_grad + forward + loss + backward are replayed from one hipGraph when
    capture succeeds (falls back to eager launches otherwise); the fused Adam follows eagerly.

    Data-parallel schedules (world > 1; `schedule`, default from PNPP_DP_SCHEDULE / PNPP_NO_OVERLAP, else "overlap"):
      captured_overlap  ONE graph: forward, backward of head + sa3, [fork] all-reduce of that 94 % of the gradient bytes beside the
                        backward pass of sa2 / sa1, all-reduce of the rest, [join].  RCCL's launches are graph nodes: no host
                        launch and no graph boundary between backward and the collective.  Needs a capturable backend (RCCL).
      captured_single   ONE graph: forward, backward, one all-reduce of the whole flat gradient (no fork / join pair).
      overlap           two graphs with the first all-reduce issued from the host between them (any backend).
      single            one graph, then one all-reduce from the host (any backend)."""
    from pnpp_hip import ops, dist as pdist

    fused_tail = os.environ.get("PNPP_FUSED_TAIL", "1") != "0"

    # the centres of step t+1 are drawn by step t's tail launch, in the 255 CUs its single workgroup leaves idle (PNPP_TAIL_SAMPLER=0:
    # every step opens with its own sampling launch); the draws and their order are the same either way
    ring = None
    if (fused_tail and os.environ.get("PNPP_TAIL_SAMPLER", "1") != "0" and model.sa1.sampler == "device" and model.sa2.sampler == "device"
            and model._presampled is None):
        from pnpp_hip import sampling as _sampling
        ring = _sampling.CentreRing(xyz.size(0), xyz.size(1), model.sa1.npoint, model.sa2.npoint, xyz.device)
        model.use_presampled(ring)
    elif fused_tail and model._presampled is not None:
        ring = model._presampled

    def tail(f, m, k):
        # fc3 + head + KL + .mean() + the seed of loss.backward() (train_single_peak_vonMises_KL.py:82-84) + fc3's backward in ONE
        # launch (12.5 us; as three launches -- fc3, head/KL, fc3 backward: 16.1 us, PNPP_FUSED_TAIL=0)
        if fused_tail:
            return ops.vm_fc_head_kl_loss_backward(f, model.fc3, m, k, next_centres=ring.job() if ring is not None else None)
        return ops.vm_head_kl_loss_backward(ops.fc_block(f, model.fc3, training=model.training), m, k)

    def loss_fn(x, m, k):
        return tail(model.trunk(x), m, k)

    def stage1(x, m, k):
        return model.levels12(x)

    def stage2(l2_xyz, l2_pts):
        _, l3 = model.sa3(l2_xyz, l2_pts)
        f = l3.view(l3.size(0), -1)
        f = ops.fc_block(f, model.fc1, model.bn1, relu=True, training=model.training)
        f = ops.fc_block(f, model.fc2, model.bn2, relu=True, dropout=model.drop, training=model.training)
        return tail(f, mu_gt, kappa_gt)

    if schedule is None:
        schedule = os.environ.get("PNPP_DP_SCHEDULE") or {"0": "overlap", "1": "single"}.get(os.environ.get("PNPP_NO_OVERLAP", ""), "overlap")
    assert schedule in DP_SCHEDULES, schedule
    dp = world > 1 and collective
    async_ar = lambda t: pdist.all_reduce_flat_grad(t, async_op=True)   # noqa: E731
    graphed, split = None, None
    if use_graph and dp and schedule in ("overlap", "captured_overlap"):
        try:
            from pnpp_hip.graph import GraphedSplitStep
            tail_off = opt.offset_of(next(model.sa3.parameters()))
            split = GraphedSplitStep(opt, stage1, stage2, [xyz, mu_gt, kappa_gt], tail_off, adopt_inputs=True,
                                     captured_all_reduce=async_ar if schedule == "captured_overlap" else None)
        except Exception as e:
            if not is_capture_error(e):   # a bug in the step itself must never be swallowed into a slower mode
                raise
            print(f"[bench] {schedule}: capture failed, trying one graph: {type(e).__name__}: {e}", file=sys.stderr)
            split = None
    if use_graph and split is None:
        try:
            from pnpp_hip.graph import GraphedStep
            # the batch is resident: no staging copy.  The Adam launch stays eager: PNPP_CAPTURED_ADAM=1 captures it too
            # (device-side step count), measured 2.5 % slower -- DESIGN.md 9
            graphed = GraphedStep(opt, loss_fn, [xyz, mu_gt, kappa_gt], adopt_inputs=True,
                                  fused_optimizer=(world == 1 and os.environ.get("PNPP_CAPTURED_ADAM") == "1"),
                                  zero_grad_in_graph=False,   # the eager Adam launch clears the gradients it has consumed
                                  captured_all_reduce=async_ar if dp and schedule == "captured_single" else None)
        except Exception as e:  # capture is an optimisation, never a requirement -- but only capture errors fall back
            if not is_capture_error(e):
                raise
            print(f"[bench] hipGraph capture failed, running eagerly: {type(e).__name__}: {e}", file=sys.stderr)
            graphed = None
    in_graph = (split is not None and split.captured_collective) or (graphed is not None and graphed._car is not None)

    def make_step(coll):
        def step():
            if split is not None:
                loss = split(xyz, mu_gt, kappa_gt, all_reduce=async_ar if coll else None)
                opt.step(grad_scale=1.0 / world)
                return loss
            if graphed is not None:
                loss = graphed(xyz, mu_gt, kappa_gt)
                if graphed.fused_optimizer:
                    return loss
            else:
                opt.zero_grad()
                loss = loss_fn(xyz, mu_gt, kappa_gt)
                if loss.requires_grad:
                    loss.backward()
            if coll and not in_graph:   # the instrumented roofline pass runs on rank 0 alone: it must not enter a collective
                pdist.all_reduce_flat_grad(opt.flat_g)
            opt.step(grad_scale=1.0 / world, zero_grad=graphed is not None)
            return loss
        return step

    mode = ("ONE hipGraph (fwd + bwd, the two all-reduces captured beside the sa2/sa1 backward pass) + eager Adam" if split is not None and in_graph
            else "two hipGraphs (fwd + sa3/head bwd | sa2/sa1 bwd) with the bucketed all-reduce overlapped + eager Adam" if split is not None
            else "ONE hipGraph (fwd + bwd + the captured all-reduce) + eager Adam" if in_graph
            else "hipGraph(fwd+loss+bwd+Adam, gradients cleared by the update)" if graphed is not None and graphed.fused_optimizer
            else "hipGraph(fwd+loss+bwd) + eager all-reduce/Adam (the update clears the gradients)" if graphed is not None else "eager")
    local = None
    if not in_graph:
        local = make_step(False)   # same launches, no collective (rank-local)
    return make_step(collective), mode, local


def kernel_cost(tag: str):
    """Algorithmic FLOPs and HBM bytes of one launch from its tag ('Kernels'): operands read once, results written
    once -- partial-sum slabs, re-reads and sector over-fetch are what `traffic` exposes against these figures."""
    def ints(pattern, text=tag):
        m = re.search(pattern, text)
        return tuple(int(x) for x in m.groups()) if m else None

    if tag.startswith("gemm_wsd3_kernel") and ",A4>" in tag:
        # the same fused product with a DENSE upstream gradient (a grouped level's middle layer): dY_l and Z_l read once
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        return 4.0 * M * N * K, 4.0 * (2.0 * M * K + 2.0 * M * N + 2.0 * K * N)
    if tag.startswith(("gemm_wsp_kernel", "gemm_wsq_kernel", "gemm_wsd3_kernel")):
        # the fused backward product of a level's last layer on gemm_wsp / gemm_wsq: dA (+ ReLU mask, sums) and dW in one launch;
        # Z_l and z_{l-1} read once, dY_{l-1} written once, the weights read and dW written once.  The kernel's own dW partial
        # slabs (one per workgroup) are NOT algorithmic bytes: they show up in `traffic` (round 3 counted them here and read 1.035 x;
        # operands-once / results-once it was 1.10 x for gemm_wsp and 1.31 x for gemm_wsq)
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        return 4.0 * M * N * K, 4.0 * (M * K + 2.0 * M * N + 2.0 * K * N)
    if tag.startswith("gemm_wsx_kernel"):
        # layer 1's backward with layer 0 folded in: dY_1 and Z_1 read once, neighbour indices; Z_0 is rebuilt (two MFMA steps per
        # tile), dY_0 is never written; weights read, dW written once (partial slabs are traffic, not algorithmic bytes)
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        return 4.0 * M * N * K + 8.0 * M * N, 4.0 * (2.0 * M * K + M + 2.0 * K * N)
    if tag.startswith(("gemm_wsf0_kernel", "gemm_wsf03_kernel")):
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        return 2.0 * M * N * K + 8.0 * M * K, 4.0 * (M * N + M + K * N)   # z_1 written once, neighbour indices read; no operand stream
    if tag.startswith("rel_moments_kernel"):
        M, = ints(r"M=(\d+)")
        return 18.0 * M, 4.0 * M                             # neighbour indices (the coordinates are L2-resident)
    if tag.startswith("xyz0_post_kernel"):
        n, s = ints(r"N=(\d+) K=64 split=(\d+)")
        return 0.0, 4.0 * n * 64.0 * (s + 1) + 2176.0 * s
    if tag.startswith("da_dw_kernel") or tag.startswith("da_dw_mid_kernel"):
        # one launch = the dA GEMM tiles and the dW blocks of a small-M backward layer (they only share dZ)
        M, = ints(r"M=(\d+)")
        n1, k1 = ints(r"dA N=(\d+) K=(\d+)")
        n2, k2 = ints(r"dW N=(\d+) K=(\d+)")
        e, = ints(r"<E(\d)")
        flops = 2.0 * M * (n1 * k1 + n2 * k2)
        byts = 4.0 * (M * k1 + k1 * n1 + M * n1 * (2 if e == 2 else 1) + M * k2 + n2 * k2)   # dZ, W, dA (+ its ReLU-mask operand), a2, dW once
        return flops, byts
    if tag.startswith("fc_bwd_fused_kernel"):   # dz rebuilt from dy, z and the keep-mask; dx = dz W and dW = dz^T x
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        return 4.0 * M * N * K, 9.0 * M * N + 4.0 * (2.0 * N * K + 2.0 * M * K)
    if tag.startswith("fc_dx_dw_kernel"):
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        return 4.0 * M * N * K, 4.0 * (M * N + 2.0 * N * K + 2.0 * M * K)
    if tag.startswith("pool_fwd"):
        G, K, C = ints(r"G=(\d+) K=(\d+) C=(\d+)")
        return 0.0, 4.0 * G * K * C + 8.0 * G * C          # read every z once; write the maxima and their positions
    if tag.startswith("pool_bwd"):
        G, K, C = ints(r"G=(\d+) K=(\d+) C=(\d+)")
        return 0.0, 16.0 * G * C                           # dout, arg-max, the ONE z element it points at; write dm
    if tag.startswith("knn_pair_kernel"):   # both grouped levels' searches in one launch; level 2 searches among level 1's S1 centres
        B, S1, N1, k1, S2, N2, k2 = ints(r"B=(\d+) \| S=(\d+) N=(\d+) k=(\d+) \| S=(\d+) N=(\d+) k=(\d+)")
        return 0.0, float(B) * (12.0 * N1 + 12.0 * S1 + 4.0 * S1 * k1 + 12.0 * N2 + 12.0 * S2 + 4.0 * S2 * k2)
    if tag.startswith("knn_kernel"):
        B, S, N, k = ints(r"B=(\d+) S=(\d+) N=(\d+) k=(\d+)")
        return 0.0, float(B) * (12.0 * N + 12.0 * S + 4.0 * S * k)
    if tag.startswith("scatter_dz_kernel"):
        B, N, C, Mc = ints(r"B=(\d+) N=(\d+) C=(\d+) M=(\d+)")
        return 0.0, 8.0 * C * B * Mc + 4.0 * B * N * C + 4.0 * B * Mc   # dY and Z of every grouped row, G per source point, indices
    if tag.startswith("gather_rel_stats_kernel"):
        M, C = ints(r"M=(\d+) C=(\d+)")
        return 6.0 * M * C, 4.0 * M * C + 4.0 * M           # Z written once, indices (P and the coordinates are L2-resident)
    if tag.startswith("dw_xyz_kernel"):
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        dzm, = ints(r"<A(\d),")
        return 2.0 * M * N * K, 4.0 * M * N * (2 if dzm == 4 else 1) + 4.0 * M
    if tag.startswith("attention_"):
        # models/point_transformer.py's self-attention (head width 16, csrc/transformer_kernels.hip ATT_DH): 2 flops per MAC of
        # the N x N products each pass forms (forward: S, PV; dQ pass: S, dP, dQ; dK/dV pass: S, dP, dV, dK); q, k, v (and o, do, the
        # row statistics in the backward passes) read once, the result written once -- the N x N matrices never leave the chip
        bnh = ints(r"B=(\d+) N=(\d+) H=(\d+)")
        if not bnh:
            return None
        B, N, H = bnh
        pair, row = 2.0 * B * H * N * N * 16.0, 4.0 * B * N * H * 16.0
        if tag.startswith("attention_fwd_kernel"):
            return 2.0 * pair, 4.0 * row + 4.0 * B * N * H
        if tag.startswith("attention_bwd_dq_kernel"):
            return 3.0 * pair, 6.0 * row + 8.0 * B * N * H
        if tag.startswith("attention_bwd_dkv_kernel"):
            return 4.0 * pair, 7.0 * row + 8.0 * B * N * H
        return None
    if tag.startswith("adam_kernel") or tag.startswith("adam"):
        n = ints(r"n=(\d+)")
        return (0.0, 28.0 * n[0]) if n else None            # p, g, m, v read; p, m, v written
    mnk = ints(r"M=(\d+) N=(\d+) K=(\d+)")
    if not mnk:
        return None
    M, N, K = mnk
    flops = 2.0 * M * N * K
    if tag.startswith(("gemm_kernel", "gemm_ws_kernel", "gemm_wsb_kernel", "gemm_wsf_kernel", "gemm_wsf3_kernel", "gemm_smallm_kernel",
                       "gemm_mid_kernel", "gemm_mid3_kernel")):
        a, e = ints(r"A(\d),E(\d)")
        byts = 4.0 * (M * N + K * N)                      # write C, read weights
        byts += 4.0 * M * K * (2 if a == 4 else 1)        # read A (dy and z for the BatchNorm-backward operand; A5: z only,
                                                           # the pooled gradient it is rebuilt from is G x C and L2-resident)
        if a == 2:                                         # gathered operand: indices, not rows, are the compulsory part
            byts = 4.0 * (M * N + K * N) + 4.0 * M
        if e == 2:
            byts += 4.0 * M * N                            # read the previous layer's z for the ReLU mask
        if ",dW>" in tag:                                  # fused weight gradient: second GEMM on the tiles already in LDS;
            flops *= 2.0                                   # dW written once (the per-worker partials are traffic, not algorithmic)
            byts += 4.0 * K * N
        return flops, byts
    if tag.startswith("dw_kernel") or tag.startswith("dw_lds_kernel"):
        a2 = ints(r",A(\d)>")
        dzm = ints(r"<A(\d),")
        byts = 4.0 * M * N * (2 if dzm and dzm[0] == 4 else 1) + 4.0 * M * K * (0 if a2 and a2[0] == 2 else 1) + 4.0 * N * K
        return flops, byts
    return None


def csrc_sha() -> str:
    """sha256 over the kernel sources (what a PMC measurement is valid for)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def dominant_roofline(rows, nsteps):
    """The `roofline` object of the bench line for the costliest kernel that has a cost model.  rows: (tag, launches, total ms)
    as pnpp_profile_report gives them, over `nsteps` steps."""
    total = sum(r[2] for r in rows)
    rows = sorted(rows, key=lambda r: -r[2])
    for tag, cnt, ms in rows:                              # dominant kernel that has a cost model
        cost = kernel_cost(tag)
        if cost is None:
            continue
        flops, byts = cost
        sec = ms * 1e-3 / cnt
        peak_tf = (MFMA_BF16_PEAK_TFLOPS if tag.startswith("gemm_wsb_kernel") else MFMA_SPLIT_PEAK_TFLOPS if tag.startswith(SPLIT_KERNELS)
                   else MFMA_F32_PEAK_TFLOPS)
        t_mfma, t_hbm = flops / (peak_tf * 1e12), byts / (HBM_PEAK_GBS * 1e9)
        if t_mfma >= t_hbm:
            ach = flops / sec / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf}
        else:
            ach = byts / sec / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        roof.update(kernel=tag, avg_us=sec * 1e6, launches_per_step=cnt / nsteps, share_of_kernel_time=ms / total,
                    alg_flops=flops, alg_bytes=byts, traffic=None)
        # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/summarize_rocprof.py), keyed by the
        # kernel instantiation and its grid
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        key = re.sub(r" M=\d+ N=\d+ K=\d+( split=\d+)?", "", tag)
        g = re.search(r"grid=(\d+)x(\d+)", key)
        if g:
            key = key[:g.start()] + f"grid={int(g.group(1)) * int(g.group(2))}"
        if os.path.exists(pmc):
            try:
                doc = json.load(open(pmc))
                meta = doc.get("_measured_at", {})
                # the counters belong to the kernel sources they were taken with: a later edit of csrc/ voids them
                if meta.get("csrc_sha256") == csrc_sha():
                    roof["traffic"] = doc.get(key, {}).get("hbm_bytes_per_launch")
                    roof["traffic_measured_at"] = meta.get("git")
                else:
                    roof["traffic_stale"] = f"profiles/pmc_traffic.json was taken at {meta.get('git')} with other kernel sources"
            except Exception:
                pass
        return roof
    return None


def roofline_leg(step, nsteps=5):
    """Instrumented pass: HIP events around every launch, aggregated per (kernel, shape) tag."""
    from pnpp_hip import _lib
    import ctypes
    lib = _lib.lib()
    torch.cuda.synchronize()
    lib.pnpp_profile_enable(1)
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    ntag = lib.pnpp_profile_report(buf, len(buf))
    lib.pnpp_profile_enable(0)
    rows = []
    for line in buf.value.decode().splitlines():
        tag, cnt, ms = line.split("\t")
        rows.append((tag, int(cnt), float(ms)))
    if ntag <= 0 or not rows:
        return None, [], None
    rows.sort(key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    table = [{"kernel": t, "launches": c, "avg_us": 1e3 * ms / c, "share": ms / total} for t, c, ms in rows[:12]]
    dump = os.environ.get("PNPP_BENCH_DUMP")
    if dump:                                               # full per-kernel table for offline analysis
        with open(dump, "w") as f:
            f.write(f"# total kernel ms per step {total / nsteps:.4f}\n")
            for t, c, ms in rows:
                f.write(f"{1e3 * ms / nsteps:9.1f} us/step  {c / nsteps:4.1f} x {1e3 * ms / c:8.1f} us  {t}\n")
    return dominant_roofline(rows, nsteps), table, total / nsteps


def cpu_baseline(B, budget_s=16.0):
    """The CPU oracle's float32 step on the host cores, two flavours of the same function (tests/test_oracle_golden.py holds them
    equal): `value` = the step through the STOCK ATen ops the reference's own modules call (F.conv2d 1x1, F.batch_norm in training
    mode, dist.topk: models/pointnet_pp_8dir.py:21-43, models/base.py:20-35) -- the representative baseline; `restatement` = the
    dtype-generic restatement the parity tests use (BatchNorm spelled as elementwise tensor expressions: slower on a CPU)."""
    from oracle import restatement as R
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    # 32 threads measured best on the GPU box's host share (8: 29, 16: 30, 32: 33, 64: 20, all 128: 9 clouds/s)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    torch.manual_seed(42)
    state = PointNetPPVonMises().state_dict()
    xyz, mu_gt, kappa_gt, _ = R.synthetic_clouds(B, N_POINTS, seed=1234)
    mask_gen = torch.Generator().manual_seed(1)

    def make_step(aten):
        P = R.cast_params(state, torch.float32)
        opt = torch.optim.Adam([v for v in P.values() if v.requires_grad], lr=1e-3)

        def step():
            opt.zero_grad()
            centres = R.replay_centres(B)
            mask = (torch.rand(B, 256, generator=mask_gen) < 0.5).float()
            if aten:
                mu, kappa = R.vonmises_forward_aten(xyz, P, centres, mask)
            else:
                mu, kappa = R.vonmises_forward(xyz, P, centres, mask, True, None)
            loss = R.kl_single(mu, kappa, mu_gt, kappa_gt).mean()
            loss.backward()
            opt.step()
            return float(loss.detach())
        return step

    def median_step(step, budget, cap):
        step()
        t0 = time.perf_counter()
        times = []
        while time.perf_counter() - t0 < budget and len(times) < cap:
            t = time.perf_counter()
            step()
            times.append(time.perf_counter() - t)
        times.sort()
        return times[len(times) // 2], len(times)

    threads = torch.get_num_threads()
    aten, rest = make_step(True), make_step(False)
    med_a, n_a = median_step(aten, budget_s, 40)
    med_r, n_r = median_step(rest, min(budget_s, 10.0), 20)
    torch.set_num_threads(1)                               # also the single-thread figure
    med1, n1 = median_step(aten, min(budget_s, 8.0), 3)
    torch.set_num_threads(threads)
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": B / med_a, "unit": "clouds/s", "cores": threads, "kind": "port",
            "sample": f"{n_a} steps of batch {B} x {N_POINTS} points, median step {med_a * 1e3:.1f} ms, fwd+loss+bwd+Adam, float32, "
                      f"oracle/restatement.py::vonmises_forward_aten (stock ATen conv2d / batch_norm / topk, the ops the reference's modules call)",
            "restatement": {"value": B / med_r, "cores": threads, "kind": "port",
                            "sample": f"{n_r} steps, median {med_r * 1e3:.1f} ms, oracle/restatement.py::vonmises_forward (elementwise BatchNorm, "
                                      f"matmul convolution: the dtype-generic form the parity tests evaluate in float64)"},
            "one_thread": {"value": B / med1, "cores": 1, "sample": f"{n1} steps of the ATen-op flavour, median {med1 * 1e3:.0f} ms"},
            "cpu_model": cpu_model, "host_cpus": os.cpu_count(), "torch": torch.__version__}


def rccl_tuning_log_setup():
    """Ask RCCL to write which algorithm / protocol it picks for each message size to a per-process FILE (never stdout: the
    one JSON line lives there).  Must run before the communicator is created.  OFF by default (PNPP_RCCL_TUNING_LOG=1 turns it on):
    the TUNING subsystem logs a line on the host for every eagerly enqueued collective -- inside the timed loop of the host-issued
    schedules -- so a run that carries it is a diagnostic run, and its line says so (`rccl.logging_perturbs_timing`)."""
    if os.environ.get("PNPP_RCCL_TUNING_LOG", "0") != "1" or "NCCL_DEBUG" in os.environ:
        return None
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"pnpp_rccl_{os.getpid()}.log")
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,TUNING", NCCL_DEBUG_FILE=path)
    return path


def rccl_choice(path):
    """Algorithm / protocol RCCL chose for the gradient messages, from its own tuning log (None under gloo)."""
    if not path:
        return None
    if not os.path.exists(path):
        return None
    algos = {0: "tree", 1: "ring", 2: "collnet_direct", 3: "collnet_chain", 4: "nvls", 5: "nvls_tree", 6: "pat"}
    protos = {0: "LL", 1: "LL128", 2: "simple"}
    seen_, chans = {}, None
    try:
        for ln in open(path, errors="replace"):
            m = re.search(r"AllReduce: (\d+) Bytes -> Algo (\d+) proto (\d+)", ln)
            if m:
                nbytes, a, pr = (int(x) for x in m.groups())
                seen_[nbytes] = {"bytes": nbytes, "algo": algos.get(a, str(a)), "proto": protos.get(pr, str(pr))}
            m = re.search(r"(\d+) coll channels", ln)
            if m:
                chans = int(m.group(1))
        os.remove(path)
    except OSError:
        return None
    big = sorted(seen_.values(), key=lambda d: -d["bytes"])[:3]
    return {"allreduce": big, "coll_channels": chans, "logging_perturbs_timing": True} if big else None


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves.  Runs in a parent that has made
    NO GPU call (importing torch makes none); the ranks are fresh child processes, the parent only waits for them."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def timed(step, n, fence):
    fence()
    t0 = time.perf_counter()
    loss = None
    for _ in range(n):
        loss = step()
    fence()
    return time.perf_counter() - t0, loss


def measure_schedules(args, world, ctrl, trial, fence, want_captured):
    """The order that cannot lose the measurement (world > 1, no schedule pinned):
      1. the host-issued schedules are built and timed; the fastest is warmed up and TIMED for exactly --steps steps: that number
         is in hand before anything untested runs;
      2. only then, and only when asked for (PNPP_DP_CAPTURED=1 on a capturable backend), the captured schedules are tried, each
         under a bounded wait; one that beats the best host-issued trial is warmed up and timed the same way (bounded as well) and
         replaces the result;
      3. if a captured schedule left the device or the communicator unusable (`trial.poisoned`), the caller prints the line of
         step 1 and leaves without touching the GPU again.
    Returns {"name", "elapsed" (max over ranks, s), "loss", "cand"}."""
    trial.safe(SAFE_SCHEDULES)
    if not trial.trial_ms:   # stream capture itself does not work here: a number from eager launches rather than no number
        trial.safe(("eager",))
    assert trial.trial_ms, f"no data-parallel schedule could be built: {trial.failed}"

    def run(name, bounded):
        cand = trial.built[name]

        def body():
            for _ in range(args.warmup):
                cand[0]()
            return timed(cand[0], args.steps, fence)
        if not bounded:
            el, loss = body()
        else:
            status, val = run_bounded(body, trial.bounded_s + 0.01 * (args.steps + args.warmup), trial.thread_setup)
            if not ctrl.all_ok(status == "ok"):
                trial.failed[name] = "its timed run " + ("timed out" if status == "timeout" else f"failed: {val}"[:160])
                trial.trial_ms.pop(name, None)
                trial.poisoned = True
                return None
            el, loss = val
        return {"name": name, "elapsed": ctrl.max(el), "loss": loss, "cand": cand}

    result = run(trial.best(SAFE_SCHEDULES + ("eager",)), bounded=False)
    if want_captured:
        trial.optional(CAPTURED_SCHEDULES)
        best = trial.best()
        if not trial.poisoned and best in CAPTURED_SCHEDULES:
            better = run(best, bounded=True)
            if better is not None and better["elapsed"] < result["elapsed"]:
                result = better
    return result


def pdist_rank():
    import torch.distributed as tdist
    return tdist.get_rank() if tdist.is_initialized() else 0


class _RehearsalStep:
    """Stand-in for a captured step in --rehearse: the flat all-reduce over gloo, in the schedule's shape, or an injected failure."""

    def __init__(self, name, flat_g, inject):
        import torch.distributed as tdist
        if inject and "@" in inject:                        # "hang@1": on rank 1 only (its peers then wait for it in the collective)
            inject, only = inject.split("@")
            inject = inject if int(only) == pdist_rank() else None
        self.name, self.g, self.inject, self.tdist = name, flat_g, inject, tdist
        if inject == "raise_build":
            raise RuntimeError("HIP error: an illegal memory access was encountered (injected by --rehearse-inject)")
        if inject == "capture_error":
            raise RuntimeError("operation not permitted when stream is capturing (injected by --rehearse-inject)")

    def __call__(self):
        if self.inject == "hang":
            time.sleep(3600)
        if self.inject == "raise":
            raise RuntimeError("NCCL error: unhandled system error (injected by --rehearse-inject)")
        from pnpp_hip import dist as pdist
        if "overlap" in self.name:
            cut = self.g.numel() // 16
            pdist.all_reduce_flat_grad(self.g[cut:])
            pdist.all_reduce_flat_grad(self.g[:cut])
        else:
            pdist.all_reduce_flat_grad(self.g)
        self.g.fill_(1.0)
        return torch.zeros(())


def rehearse(args):
    """Launcher / rendezvous / collective plumbing and the WHOLE schedule-selection flow WITHOUT kernels (`--rehearse`, CPU + gloo):
    what tests/ can run in a container that has no GPU.  The line it prints is marked as a rehearsal and carries no throughput.
    --rehearse-inject captured_single=raise,captured_overlap=hang (raise | raise_build | capture_error | hang) makes a captured
    schedule fail the way a real one could; the line must still come out, with the host-issued candidates' times."""
    import torch.distributed as tdist
    from pnpp_hip import dist as pdist
    rank, _, world = pdist.init_from_env(backend="gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    flat_g = torch.ones(1_465_922)                          # the flat gradient of PointNetPPVonMises
    seen = torch.ones(1)
    if world > 1:
        tdist.all_reduce(seen)
    pdist.all_reduce_flat_grad(flat_g)
    allreduce_ok = bool(torch.all(flat_g == float(world)))
    inject = dict(kv.split("=") for kv in args.rehearse_inject.split(",") if kv)
    ctrl = Ctrl(world)

    def fence():
        if world > 1:
            tdist.barrier()

    def build(name):
        return _RehearsalStep(name, flat_g, inject.get(name)), SCHEDULE_MODE_PREFIX[name], None

    trial = ScheduleTrial(build, lambda step, n: timed(step, n, fence)[0], ctrl, bounded_s=float(os.environ.get("PNPP_DP_BOUNDED_S", "60")))
    res = measure_schedules(args, world, ctrl, trial, fence, want_captured=bool(inject) or os.environ.get("PNPP_DP_CAPTURED") == "1")
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL of the launcher, the schedule selection and the flat-gradient all-reduce (gloo, no kernels, not a measurement)",
                          "value": None, "unit": "clouds/s", "n_gpus": world, "n_ranks_seen": int(seen.item()),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * res["elapsed"] / max(args.steps, 1),
                          "rehearsal": True, "allreduce_ok": allreduce_ok, "config": {"dp_schedule": trial.report(res["name"])},
                          "poisoned": trial.poisoned}), flush=True)
    if trial.poisoned:                                      # a helper thread still hangs: leave without waiting for anything
        sys.stderr.flush()
        os._exit(0)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU (config 2: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, the reference's arithmetic: the headline) or bf16 (opt-in throughput mode: bf16 MFMA operands, "
                         "f32 accumulate, reported under its own metric key)")
    ap.add_argument("--f32-products", choices=["split", "mfma"], default="split",
                    help="how the float32 products of the large GEMMs are formed: split (default: six exact bf16 x bf16 partial products of "
                         "three-way operand splits on the bf16 matrix pipe, float32 accumulate -- float32 results to float32 rounding) or "
                         "mfma (v_mfma_f32_32x32x2_f32)")
    ap.add_argument("--no-mfma-variant", action="store_true",
                    help="skip the second line measured with --f32-products mfma (N = 1, float32 only; reported as f32_mfma_variant)")
    ap.add_argument("--bf16-variant", action="store_true",
                    help="also measure the secondary bf16-operand line (never the headline; off by default since round 4)")
    ap.add_argument("--no-bf16-variant", action="store_true", help="accepted for older command lines: the variant is off by default")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU/gloo rehearsal of the launcher, the schedule selection and the collective plumbing only (no kernels)")
    ap.add_argument("--rehearse-inject", default="", help="--rehearse only: name=raise|raise_build|capture_error|hang[@rank][,name=...]")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # no launcher around us: become one (no GPU call made so far)
        sys.exit(self_launch(args))
    if args.rehearse:
        return rehearse(args)

    from pnpp_hip import _lib, dist as pdist, optim, ops
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    import synthetic
    import torch.distributed as tdist

    _lib.lib()                                             # fail loudly if the HIP extension is missing
    ops.set_matmul_precision(args.precision)
    ops.set_float32_products(args.f32_products)
    rccl_log = rccl_tuning_log_setup() if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None
    rank, local_rank, world = pdist.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", torch.cuda.current_device())

    torch.manual_seed(42)                                  # train_single_peak_vonMises_KL.py:19-20
    model = PointNetPPVonMises(sampler="device").to(dev).train()   # centre sampling on the GPU (same distribution as randperm)
    opt = optim.FlatAdam(model.parameters(), lr=1e-3)
    pdist.broadcast_flat(opt.flat_p)
    B = args.batch
    xyz, mu_gt, kappa_gt, _ = synthetic.rotated_clouds(B, N_POINTS, seed=1234 + rank)
    xyz, mu_gt, kappa_gt = xyz.to(dev), mu_gt.to(dev), kappa_gt.to(dev)
    seen = torch.ones(1, device=dev)
    if world > 1:
        tdist.all_reduce(seen)                             # every rank really is in the job (and the communicator exists)
    ctrl = Ctrl(world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    def emit(elapsed, final_loss, launch_mode, dp_schedule, exposed_us=None, replicas=None, bf16=None, roof=None, table=(), kernel_ms=None,
             cpu=None, note=None, mfma=None):
        ms = 1e3 * elapsed / args.steps
        per_gpu = B * args.steps / elapsed
        out = {
            "metric": "clouds/sec fwd+bwd, pointnet_pp_vonMises N=1024" + (
                "" if args.precision == "f32" else ", bf16-operand MFMA variant (f32 accumulate)"), "value": world * per_gpu,
            "unit": "clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "configs[1]: models/pointnet_pp_vonMises.py single-peak KL, N=1024, batch=32 per GPU, "
                                   "fwd+loss+bwd+allreduce+Adam, random-init weights (seed 42), device-side centre sampling",
                       "float32_products": ("six exact bf16 x bf16 partial products of three-way operand splits on the bf16 matrix pipe, float32 "
                                            "accumulate (float32 results to float32 rounding; same parity gates as the float32 MFMA form: "
                                            "tests/test_gpu_split_products.py)" if args.f32_products == "split" and args.precision == "f32"
                                            else "v_mfma_f32_32x32x2_f32" if args.precision == "f32" else "n/a (bf16 operands)"),
                       "per_gpu_batch": B, "global_batch": B * world, "points": N_POINTS,
                       "parallelism": f"dp{world}" if world > 1 else "single", "launch": launch_mode,
                       **({"dp_schedule": dp_schedule} if dp_schedule else {})},
            "final_loss": final_loss, "n_ranks_seen": int(seen_host), "allreduce_exposed_us": exposed_us,
            **({"replicas_rel_spread": replicas} if world > 1 else {}),
            **({"rccl": rccl_choice(rccl_log)} if world > 1 and rccl_log else {}),
            # whole step against both roofs: algorithmic FLOPs / bytes per cloud x clouds/s per GPU
            "mfma_fraction": per_gpu * FLOPS_PER_CLOUD / (MFMA_F32_PEAK_TFLOPS * 1e12),
            "hbm_fraction": per_gpu * BYTES_PER_CLOUD / (HBM_PEAK_GBS * 1e9),
            "kernel_ms_per_step": kernel_ms, **({"f32_mfma_variant": mfma} if mfma else {}), **({"bf16_variant": bf16} if bf16 else {}),
            **({"note": note} if note else {}),
            "roofline": roof, "cpu_baseline": cpu, "top_kernels": list(table),
        }
        print(json.dumps(out), flush=True)

    seen_host = float(seen.item())
    forced = os.environ.get("PNPP_DP_SCHEDULE") or {"0": "overlap", "1": "single"}.get(os.environ.get("PNPP_NO_OVERLAP", ""))
    dp_schedule = None
    if world > 1 and not args.no_graph and forced is None:
        # Data-parallel schedule, chosen by measurement (never inside the timed region).  Overlap is not free on this chip: the
        # collective's workgroups need CUs that the persistent GEMM kernels of the backward pass assume to own -- so every rank builds
        # the host-issued schedules, times them (MAX over ranks) and all take the fastest; PNPP_DP_SCHEDULE (or PNPP_NO_OVERLAP=0/1)
        # pins one, PNPP_DP_CAPTURED=1 adds the captured ones behind the secured measurement (measure_schedules).
        capturable = tdist.get_backend() == "nccl"         # RCCL launches are stream work; gloo goes through the host
        want_captured = os.environ.get("PNPP_DP_CAPTURED") == "1"
        trial = ScheduleTrial(lambda name: build_step(model, opt, xyz, mu_gt, kappa_gt, world, name != "eager",
                                                      schedule=None if name == "eager" else name),
                              lambda step, n: timed(step, n, fence)[0], ctrl,
                              bounded_s=float(os.environ.get("PNPP_DP_BOUNDED_S", "60")),
                              thread_setup=lambda: torch.cuda.set_device(dev))
        if not want_captured:
            trial.failed.update({n: "opt-in (PNPP_DP_CAPTURED=1): never run with more than one RCCL rank" for n in CAPTURED_SCHEDULES})
        elif not capturable:
            trial.failed.update({n: f"backend {tdist.get_backend()} cannot be captured" for n in CAPTURED_SCHEDULES})
        res = measure_schedules(args, world, ctrl, trial, fence, want_captured and capturable)
        dp_schedule = trial.report(res["name"])
        elapsed, loss = res["elapsed"], res["loss"]
        step, launch_mode, step_local = res["cand"]
        if trial.poisoned:
            # a captured schedule hung or faulted AFTER the host-issued measurement was taken: print that measurement and leave.  No
            # further GPU call, no collective on the communicator that hung, no wait for the helper thread; a rank never re-execs.
            if rank == 0:
                emit(elapsed, None, launch_mode, dp_schedule,
                     note="a captured schedule left the device or the communicator unusable; this is the host-issued measurement taken before it")
            sys.stderr.flush()
            os._exit(0)
        if step_local is None:                             # captured collective: its collective-free twin is the plain graph
            step_local = build_step(model, opt, xyz, mu_gt, kappa_gt, world, True, collective=False)[0]
    else:
        step, launch_mode, step_local = build_step(model, opt, xyz, mu_gt, kappa_gt, world, not args.no_graph, schedule=forced)
        if step_local is None:
            step_local = build_step(model, opt, xyz, mu_gt, kappa_gt, world, not args.no_graph, collective=False)[0]
        for _ in range(args.warmup):
            step()
        el, loss = timed(step, args.steps, fence)
        elapsed = ctrl.max(el)
    final_loss = float(loss.detach())
    eager_step, _, _ = build_step(model, opt, xyz, mu_gt, kappa_gt, world, False, collective=False)   # rank-local, for the roofline pass

    exposed_us, replicas = None, None
    if world > 1:
        # every replica took the same reduced gradients: their parameters must still be the same numbers (a collective that ran in
        # the wrong place of a schedule shows up here, not in the timing)
        chk = opt.flat_p.double().abs().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        tdist.all_reduce(lo, op=tdist.ReduceOp.MIN)
        tdist.all_reduce(hi, op=tdist.ReduceOp.MAX)
        replicas = float((hi - lo) / hi.clamp_min(1e-30))
        # the same captured step without the collective: what the all-reduce costs beyond what backward hides
        for _ in range(min(args.warmup, 5)):
            step_local()
        el2, _ = timed(step_local, args.steps, fence)
        exposed_us = 1e6 * (elapsed - ctrl.max(el2)) / args.steps
        pdist.broadcast_flat(opt.flat_p)                   # replicas drifted apart in the local-only steps: not used after this

    # the same step with the float32 products on v_mfma_f32_32x32x2_f32 (what rounds 1-3 measured): beside the headline, so that what
    # the split products buy -- and that they change nothing else -- is read off one run
    mfma = None
    if args.precision == "f32" and args.f32_products == "split" and world == 1 and not args.no_mfma_variant:
        ops.set_float32_products("mfma")
        torch.manual_seed(42)
        mm = PointNetPPVonMises(sampler="device").to(dev).train()
        om = optim.FlatAdam(mm.parameters(), lr=1e-3)
        stepm, modem, _ = build_step(mm, om, xyz, mu_gt, kappa_gt, 1, not args.no_graph)
        for _ in range(args.warmup):
            stepm()
        nm = max(20, args.steps // 2)
        elm, lossm = timed(stepm, nm, fence)
        ops.set_float32_products("split")
        mfma = {"metric": "clouds/sec fwd+bwd, pointnet_pp_vonMises N=1024, float32 products on v_mfma_f32_32x32x2_f32",
                "value": B * nm / elm, "unit": "clouds/s", "ms_per_step": 1e3 * elm / nm, "steps": nm, "dtype": "f32",
                "final_loss": float(lossm.detach()), "launch": modem}
        del stepm, mm, om

    # secondary line (never the headline, opt-in): the same step with bf16 MFMA operands in the grouped layers' large GEMMs
    bf16 = None
    if args.precision == "f32" and world == 1 and args.bf16_variant and not args.no_bf16_variant:
        ops.set_matmul_precision("bf16")
        torch.manual_seed(42)
        m16 = PointNetPPVonMises(sampler="device").to(dev).train()
        o16 = optim.FlatAdam(m16.parameters(), lr=1e-3)
        step16, mode16, _ = build_step(m16, o16, xyz, mu_gt, kappa_gt, 1, not args.no_graph)
        for _ in range(args.warmup):
            step16()
        n16 = max(20, args.steps // 2)
        el16, loss16 = timed(step16, n16, fence)
        ops.set_matmul_precision("f32")
        bf16 = {"metric": "clouds/sec fwd+bwd, pointnet_pp_vonMises N=1024, bf16-operand MFMA variant (f32 accumulate)",
                "value": B * n16 / el16, "unit": "clouds/s", "ms_per_step": 1e3 * el16 / n16, "steps": n16, "dtype": "bf16",
                "final_loss": float(loss16.detach()), "launch": mode16,
                "tolerance": "tests/test_gpu_bf16.py: bit-exact on bf16-representable data; vs fp64 at B=32 |dloss| 8e-2 (1.5 %), flat gradient relL2 0.44"}
        del step16, m16, o16

    roof, table, kernel_ms = (None, [], None)
    if not args.no_roofline and rank == 0:
        roof, table, kernel_ms = roofline_leg(eager_step)  # per-launch events need individual launches, not a graph replay
    cpu = None
    if world > 1:
        tdist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(B)
        emit(elapsed, final_loss, launch_mode, dp_schedule, exposed_us, replicas, bf16, roof, table, kernel_ms, cpu, mfma=mfma)
    elif rccl_log and os.path.exists(rccl_log):
        os.remove(rccl_log)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
