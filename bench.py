"""bench.py -- QA-pairs/sec of the CRCT co-attention training step on MI355X.

One "step" = forward + joint CE/L1 loss + backward (+ RCCL gradient all-reduce, overlapped, and the 9-float stats
all-reduce of train.py:181-189 when N > 1) + fused AdamW + LR-schedule step, on one batch of synthetic PlotQA-shaped
inputs, dropout ENABLED (p = 0.1), config ``config/vilbert.json`` with ``v_feature_size = 2048`` (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks (one fresh process per GPU, as train.py:356-363 does)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # ... or under a launcher that has set RANK / WORLD_SIZE already

Prints ONE JSON line on rank 0:
  value        : whole-job QA-pairs/s with the batches already resident in HBM when the timed region starts (a pool of 8
                 pre-staged batches is rotated).
  config.h2d_inclusive : the same step fed from PAGEABLE HOST batches through crct.input_pipeline.DevicePrefetcher (one
                 pinned staging buffer and ONE async copy per batch on a copy stream, double-buffered; features shipped
                 as bf16 unless --host-feat fp32), measured in the same run -- the PCIe-inclusive rate.
  roofline     : the FFN GEMM group the north star names: text / visual FFN-up and FFN-down, forward and
                 data gradient (each its own kernel launch).  achieved = sum of their algorithmic FLOPs / sum of their kernel
                 durations, measured live IN the step: during --profile-steps extra steps every GEMM kernel is dispatched with
                 a start / stop event pair (hipExtLaunchKernelGGL), i.e. the begin / end stamps of the kernel itself -- the
                 quantity ``rocprofv3 --kernel-trace --stats`` averages -- and the engine tags every launch with its model site
                 (CrctGemmArgs.site).  roofline.ffn lists the per-site figures (``*_wgrad``: the layer's weight gradients are
                 ONE grouped kernel, its duration is apportioned by FLOP share); config.gemm_sites has every site of the step,
                 config.gemm_variants the kernel configurations.  peak = 2.5 PFLOP/s dense bf16 MFMA (also for the non-scaled
                 fp8 MFMA, which runs at the bf16 rate).  traffic = fabric-side bytes per launch of the group from the
                 committed PMC passes (profiles/r*_pmc_sites.json), used only while the kernel sources hash to what was measured.
  cpu_baseline : the CPU oracle (fp32 PyTorch restatement, ``kind: port``) doing forward+backward of the same batch
                 shape on this node's host cores (rank 0, N = 1 only, bounded sample).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402

from crct import config as CFG            # noqa: E402
from crct import synthetic as S           # noqa: E402
from crct import lib as L                 # noqa: E402

PEAK_BF16_TFLOPS = 2500.0                 # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_FP8_SCALED_TFLOPS = 5000.0           # ... ~5 PF dense fp8 through the block-scaled MFMA the fp8 GEMMs issue (the plain fp8 MFMA runs at the bf16 rate)
FLOP_PER_QA = {(36, 20, 2048): 32.9e9, (100, 40, 2048): 79.8e9, (44, 124, 1024): 121.2e9}   # SURVEY.md 8a / 8d, fwd+bwd
# Named workloads (batch, visual elements, text tokens, feature width).  "baseline" = BASELINE.json configs[1], the default and the
# driver's line; "long-context" = configs[3]; "plotqa-real" = the shape the reference's own PlotQA configuration trains at
# (CRCT/config/plotqa.json:5-6 max_vis_features 44 / max_seq_len 124, every sample padded to it by CRCT/utils.py:152-160;
# config/vilbert.json v_feature_size 1024): an EXTRA line, M_t = 9 920 text rows per GEMM, 121.2 GFLOP per QA pair.
WORKLOADS = {"baseline": (80, 36, 20, 2048), "long-context": (64, 100, 40, 2048), "plotqa-real": (80, 44, 124, 1024)}
VARIANT_NAMES = {0: "fwd", 1: "dgrad", 2: "wgrad"}
TILE_NAMES = {0: "dma128x128w4s3", 1: "dma128x64w4s4", 2: "dma64x128w4s4", 3: "dma64x64w4s4", 4: "dma128x128w8s3",
              5: "dma128x256w8s3", 6: "dma256x128w8s3", 7: "dma256x128w8s2", 8: "dma128x128w8s4", 9: "dma128x128w8s2",
              10: "dma128x64w8s3", 11: "dma64x128w8s3", 12: "dma128x64w8s2", 13: "dma128x64w8s4", 14: "dma128x64w8s6", 15: "dma128x64w8s3", 16: "reg128x128", 17: "reg128x64",
              18: "reg64x128", 19: "reg64x64", 20: "f8e4m3_128x64w8s2", 21: "f8e4m3_128x64w8s3", 22: "dma160x128w4s2", 23: "dma160x128w4s3",
              24: "dma160x96w4s3", 25: "dma160x96w4s2", 26: "dma96x64w4s3", 27: "dma96x64w4s4", 28: "dma160x64w4s3", 29: "dma64x96w4s4", 30: "dma160x128w4s3p", 31: "dma160x128w4s2p",
              32: "dma128x128w4s3p", 33: "dma128x128w8s3p", 34: "dma256x128w8s3p", 35: "dma160x96w4s3p",
              36: "f8t_128x128w8s3", 37: "f8t_128x128w8s2", 46: "ldr128x64w8+4s3", 47: "ldr128x64w8+4s2",
              48: "ldr128x128w8+4s3", 49: "ldr128x128w8+4s2", 50: "ldr256x128w8+4s3", 51: "ldr128x64w8+4s4", 52: "ldr128x64w8+2s3",
              53: "ldr128x128w4+4s3", 54: "ldr128x64w4+2s3", 55: "ldr256x128w8+4s2", 56: "ldrp128x64w4+2s3", 57: "ldrp128x64w8+4s3",
              58: "ldrp128x128w4+4s3", 59: "ldrp128x128w8+4s3", 61: "ldrp128x64w4+2s4", 62: "ldrp128x128w4+4s2",
              63: "ldrp128x64w4+4s3", 64: "ldrp128x64w4+2s2", 60: "ldr256x128w4+4s3", 65: "ldr256x128w4+4s2",
              66: "ldrh256x128w4+4s3", 67: "ldrh256x128w8+4s3", 68: "ldrh128x128w4+4s3", 69: "ldrh128x64w4+2s3", 70: "ldrh128x128w8+4s3", 71: "ldrh256x128w4+4s2"}
KERNEL_SOURCES = ("cqa-crct_amd/csrc/gemm.hip", "cqa-crct_amd/csrc/engine.cpp", "cqa-crct_amd/csrc/common.hip.h")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None, help="a named shape (sets --batch / --vis / --tokens / --feat): %s"
                    % ", ".join("%s = B %d, V %d, T %d, F_v %d" % ((k,) + v) for k, v in sorted(WORKLOADS.items())))
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--vis", type=int, default=36)
    ap.add_argument("--tokens", type=int, default=20)
    ap.add_argument("--feat", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=80)
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--input", choices=("resident", "prefetch", "sync"), default="resident",
                    help="what `value` is measured on.  resident: batches already in HBM (the contract's headline); prefetch: pageable "
                         "host batches through crct.input_pipeline.DevicePrefetcher; sync: pageable host batches moved by the step "
                         "adapter's synchronous .to(device), as the reference does")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="length in seconds of the extra sustained leg (config.sustained): the same step "
                    "repeated back to back for at least this long, so that an outside sampler (rocm-smi, 5 s period) sees the GPU busy; 0 = skip")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the PCIe-inclusive leg (config.h2d_inclusive)")
    ap.add_argument("--host-feat", choices=("bf16", "fp32"), default="bf16", help="dtype of image_feat in the host batches of the prefetch legs")
    ap.add_argument("--fuse-zero-grad", action="store_true", help="AdamW zeroes the gradients it consumes (measured: no gain)")
    ap.add_argument("--eager-zero-grad", action="store_true", help="zero_grad() fills the whole gradient buffer and backward "
                    "accumulates into it (default: lazy clear + overwritten weight gradients, same results, -0.3 ms per step)")
    ap.add_argument("--adamw-wgs", type=int, default=-1, help="workgroups per overlapped AdamW launch (0 = full width; default: the optimizer's)")
    ap.add_argument("--opt-early", type=int, default=0, help="1: AdamW of a segment starts when backward has finished the segment")
    ap.add_argument("--no-opt-overlap", action="store_true", help="run AdamW as one launch on the main stream")
    ap.add_argument("--grad-dtype", choices=("bf16", "fp32"), default="bf16", help="payload of the gradient all-reduce when N > 1 (bf16: 2 B per "
                    "parameter, consumed by the fused AdamW as it lies; fp32: the reference's payload)")
    ap.add_argument("--force-exchange", action="store_true", help="N = 1 only: run the bucketed exchange (per-segment callback, pack, all_reduce on a "
                    "single-rank RCCL communicator, AdamW from the bf16 buffer) inside the timed step -- the N > 1 code path on a 1-GPU box")
    ap.add_argument("--ghost-ranks", type=int, default=0, help="with --force-exchange on ONE GPU: predict the N-rank step -- behind every bucket's "
                    "single-rank all-reduce a stand-in kernel with RCCL's footprint (--ghost-channels workgroups streaming the bucket 2 (N - 1) / N "
                    "times through HBM, held for bytes x 2 (N - 1) / N / --ghost-bus-gbps) runs on the exchange's stream; reported as "
                    "config.gradient_allreduce.ghost.  Nothing is reduced: the losses are the 1-rank run's")
    ap.add_argument("--ghost-channels", type=int, default=128, help="workgroups of the stand-in (RCCL logs its channel count at init: 128 on this stack)")
    ap.add_argument("--ghost-bus-gbps", type=float, default=350.0, help="bus bandwidth the stand-in assumes (7 xGMI links x ~153 GB/s per GPU, ~350 GB/s "
                    "reached by ring all-reduces on 8 x MI300-class nodes)")
    ap.add_argument("--emulate-ranks", type=int, default=1, help="test hook (N = 1): every step's batch is the CONCATENATION of the batches R "
                    "ranks would draw (batch R x --batch): the run a data-parallel R-rank run must agree with (config.global_loss)")
    ap.add_argument("--residual-bf16", action="store_true", help="params['residual_fp32'] = False: the residual stream stored as bf16 (rounds 1 - 5; "
                    "3 %% faster, gradient-cosine deficit against the fp32 oracle x 5: profiles/r6_residual_stream_parity.txt); not the default, "
                    "never the headline line")
    ap.add_argument("--no-dropout", action="store_true", help="test hook: dropout probabilities 0 (run-to-run and rank-count independent losses)")
    ap.add_argument("--exchange-skip", default="", help="timing experiment: comma list of exchange parts to leave out (pack, collective, stats)")
    ap.add_argument("--vis-stream", type=int, default=1, help="0: the visual stream's layers on the caller's stream (developer timing experiment)")
    ap.add_argument("--embed-scatter-split", action="store_true", help="developer A/B: the text embedding's backward sums and scatter as two launches")
    ap.add_argument("--wgrad-concat", type=int, default=-1, help="developer A/B: 0 = per-problem XCD rectangles for the grouped weight gradients (crct_gemm_group_concat)")
    ap.add_argument("--wgrad-flush", type=int, default=-1, help="developer A/B: flush points of a layer's queued weight gradients (crct_engine_set_wgrad_flush)")
    ap.add_argument("--wgrad-cfg", type=int, default=-1, help="developer A/B: kernel configuration of the grouped weight-gradient launches (crct_gemm_group_wgrad_config)")
    ap.add_argument("--wgrad-wgs", type=int, default=-1, help="cap on the workgroups of a layer's grouped weight-gradient launch (0 = one per tile)")
    ap.add_argument("--wgrad-target-wgs", type=int, default=-1, help="developer A/B: target size of the persistent grid of the engine's grouped bf16 "
                    "weight-gradient launches (crct_engine_set_wgrad_workgroups; 0 = one workgroup per tile, the engine's default is 96)")
    ap.add_argument("--wgrad-target-rows", type=int, default=3000, help="developer A/B: ... for data streams of at most this many token rows")
    ap.add_argument("--wgrad-streams", type=int, default=-1, help="weight-gradient side streams of the engine: 1 = one per data stream, 2 = ONE shared "
                    "stream (default: 1 without a gradient exchange, 2 with one -- the exchange then has a hardware queue to itself)")
    ap.add_argument("--exchange-pack-all", action="store_true", help="developer A/B: the exchange packs every gradient from fp32 (the weight-gradient "
                    "GEMMs do not write the bf16 communication buffer themselves)")
    ap.add_argument("--adamw-wide-first", type=int, default=-1, help="developer A/B: how many of the first overlapped AdamW launches run unthrottled")
    ap.add_argument("--adamw-groups", type=int, default=-1, help="developer A/B: the overlapped AdamW in this many launches (0 = one per backward segment)")
    ap.add_argument("--fp8-bf16-wgrad", action="store_true", help="--dtype fp8 with bf16 weight gradients (fp8 forward and data gradients)")
    ap.add_argument("--fp8-plain-mfma", action="store_true", help="developer A/B (--dtype fp8): the round-3 fp8 GEMMs (four v_mfma_f32_16x16x32_fp8 per "
                    "128-deep K tile) instead of one v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (crct_gemm_fp8_scaled_mfma)")
    ap.add_argument("--fp8-bf16-forward", action="store_true", help="--dtype fp8 with bf16 FORWARD GEMMs: only the data and weight gradients run on fp8 "
                    "operands (params['fp8_forward'] = False: the high-fidelity fp8 mode, gradient cosine ~0.97 against fp32 instead of ~0.87)")
    ap.add_argument("--fp8-forward-only", action="store_true", help="--dtype fp8 with the round-2 scope: fp8 forward GEMMs, bf16 backward")
    ap.add_argument("--site-policy", default="", help="developer A/B: comma list of site:kind:phase:cfg:split_k overrides of the per-site "
                    "GEMM launch policy (crct_engine_set_site_policy), e.g. t.ffn_down:fwd:0:4:3; reported in config.site_policy")
    ap.add_argument("--class-policy", default="", help="developer A/B: comma list of class=cfg overrides of the GEMM shape-class table "
                    "(crct_gemm_class_config; classes S.w S.n S.nl M.w M.n M.nl L.w L.n L.nl), e.g. L.w=9,L.n=9; reported in config.class_policy")
    ap.add_argument("--launch-log", default="", help="developer tooling: write the GEMM launch log of the timed region to this JSON file "
                    "(tools/pmc_sites.py matches it against a rocprofv3 counter collection)")
    ap.add_argument("--rank-timeout-s", type=float, default=600.0, help="`python bench.py --gpus N` as its own launcher: seconds after which the parent "
                    "ends every rank it started and exits non-zero (a rank stuck in the RCCL bootstrap would otherwise hold the job until the "
                    "driver's own limit); each rank's last stderr lines are relayed")
    ap.add_argument("--launch-check", action="store_true", help="test hook: every rank reports its rendezvous environment (rank 0 as the JSON line) and "
                    "exits before touching a GPU -- the self-launch path of `python bench.py --gpus N` checked on a box without GPUs")
    ap.add_argument("--dtype", choices=("bf16", "fp8"), default="bf16",
                    help="bf16: the headline (BASELINE configs[1]).  fp8: BASELINE configs[4] -- the QKV / FFN GEMMs of the forward pass on "
                         "OCP e4m3 operands with per-tensor delayed scaling and fp32 accumulation, backward and everything else bf16")
    a = ap.parse_args()
    if a.workload:
        a.batch, a.vis, a.tokens, a.feat = WORKLOADS[a.workload]
    return a


def stage(batch, dev):
    return {k: v.to(dev) for k, v in batch.items()}


def gemm_profile(run_step, n_steps):
    """Begin / end stamps of every GEMM kernel over `n_steps` extra steps (same process, every internal stream): rows per kernel
    configuration and rows per model site (CRCT_SITE_* x forward / dgrad / wgrad)."""
    lib = L.load()
    lib.crct_prof_reset()
    lib.crct_prof_enable(1)
    for _ in range(n_steps):
        run_step()
    torch.cuda.synchronize()
    lib.crct_prof_enable(0)
    rows, sites = [], []

    def row(label, cnt, fl, ms, **extra):
        return dict(kernel=label, launches_per_step=cnt / n_steps, gflop_per_launch=fl / cnt / 1e9, us_per_launch=ms * 1e3 / cnt,
                    ms_per_step=ms / n_steps, tflops=fl / (ms * 1e-3) / 1e12, frac_of_bf16_peak=fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, **extra)

    for v in range(225):
        cnt, fl, ms = C.c_long(), C.c_double(), C.c_double()
        if lib.crct_prof_read(v, C.byref(cnt), C.byref(fl), C.byref(ms)) != 0 or cnt.value == 0 or ms.value <= 0:
            continue
        rows.append(row("gemm<%s,%s>" % (TILE_NAMES.get(v // 3, "?"), VARIANT_NAMES[v % 3]), cnt.value, fl.value, ms.value))
    for s in range(1, len(L.SITE_NAMES)):
        for k in range(3):
            cnt, fl, ms, app = C.c_long(), C.c_double(), C.c_double(), C.c_int()
            if lib.crct_prof_read_site(s, k, C.byref(cnt), C.byref(fl), C.byref(ms), C.byref(app)) != 0 or cnt.value == 0 or ms.value <= 0:
                continue
            sites.append(row("%s.%s" % (L.SITE_NAMES[s], L.KIND_NAMES[k]), cnt.value, fl.value, ms.value, apportioned=bool(app.value)))
    lib.crct_prof_reset()
    return rows, sites


def critical_path(run_step, core, n_steps):
    """config.critical_path: what the step's launch structure looks like from inside the process.  During `n_steps` extra steps EVERY
    kernel the library launches carries a start / stop event pair (crct_prof_enable(2): hipExtLaunchKernelGGL stamps, the figures
    rocprofv3's kernel trace reports) and is remembered with its HIP stream: launches per step, kernels and busy time per stream
    (= hardware queue: the engine places its streams on distinct queues, csrc/streams.hip), and how long k kernels were in flight at
    once.  The text data stream is the step's dependent chain (282 of ~625 launches in round 4): its kernel count and busy time are
    what a fusion has to shorten.  Stock-torch kernels of the step (the output snapshot, autograd's seed) are not stamped."""
    lib = L.load()
    lib.crct_prof_reset()
    torch.cuda.synchronize()
    lib.crct_prof_enable(2)
    for _ in range(n_steps):
        run_step()
    torch.cuda.synchronize()
    lib.crct_prof_enable(0)
    n = lib.crct_prof_stamp_count()
    names = {torch.cuda.current_stream().cuda_stream or 0: "text (caller's stream)"}
    eng = core._engine
    if eng is not None:
        arr = (C.c_void_p * 4)()
        if lib.crct_engine_streams(eng.handle, arr) == 0:
            for ptr, nm in zip(arr, ("visual", "weight gradients (text)", "weight gradients (visual)", "auxiliary (AdamW / exchange)")):
                if ptr:
                    names.setdefault(ptr, nm)
    per, edges = {}, []
    st, t0, t1 = C.c_void_p(), C.c_double(), C.c_double()
    for i in range(n):
        if lib.crct_prof_stamp_read(i, C.byref(st), C.byref(t0), C.byref(t1)) != 0:
            continue
        q = per.setdefault(names.get(st.value or 0, "stream %#x" % (st.value or 0)), dict(kernels=0, busy_ms=0.0, first=t0.value, last=t1.value))
        q["kernels"] += 1
        q["busy_ms"] += t1.value - t0.value
        q["first"], q["last"] = min(q["first"], t0.value), max(q["last"], t1.value)
        edges += [(t0.value, 1), (t1.value, -1)]
    lib.crct_prof_reset()
    edges.sort()
    in_flight, depth, prev = {}, 0, None
    for t, d in edges:
        if prev is not None and t > prev:
            in_flight[depth] = in_flight.get(depth, 0.0) + (t - prev)
        depth += d
        prev = t
    span = (edges[-1][0] - edges[0][0]) if edges else 0.0
    return {"steps_profiled": n_steps, "launches_per_step": n / max(n_steps, 1),
            "queues": {k: {"kernels_per_step": v["kernels"] / n_steps, "busy_ms_per_step": v["busy_ms"] / n_steps} for k, v in sorted(per.items())},
            "ms_with_k_kernels_in_flight_per_step": {str(k): v / n_steps for k, v in sorted(in_flight.items())},
            "span_ms_per_step": span / max(n_steps, 1),
            "note": "library kernels only (every launch stamped with hipExtLaunchKernelGGL start / stop events on its own stream); the stamped "
                    "steps are host-bound (an event pair per launch), so kernels of different streams overlap less than in the timed step"}


FFN_SITES = ("t.ffn_up", "t.ffn_down", "v.ffn_up", "v.ffn_down")


def ffn_roofline(sites, n_profiled, kind_peak=None):
    """fraction of the FFN-GEMM roofline = sum of FFN GEMM FLOPs / sum of their kernel time / peak.
    kind_peak: dense MFMA peak (TFLOP/s) of the instruction each pass issues -- 2500 for bf16 (and the plain fp8 MFMA, which runs at the
    bf16 rate), 5000 where the fp8 GEMMs issue v_mfma_scale_f32_16x16x128_f8f6f4 (twice the rate; MI355X_MICROARCH.md).  A group that
    mixes instructions is priced by the time its FLOPs would take at each kernel's own peak."""
    by = {r["kernel"]: r for r in sites}
    kind_peak = kind_peak or {}
    pk = lambda k: kind_peak.get(k, PEAK_BF16_TFLOPS)      # noqa: E731

    def group(kinds):
        sel = [(by["%s.%s" % (s, k)], k) for s in FFN_SITES for k in kinds if "%s.%s" % (s, k) in by]
        if not sel:
            return None
        fl = sum(r["gflop_per_launch"] * r["launches_per_step"] for r, _ in sel)        # GFLOP per step
        ms = sum(r["ms_per_step"] for r, _ in sel)
        n = sum(r["launches_per_step"] for r, _ in sel)
        ideal_ms = sum(r["gflop_per_launch"] * r["launches_per_step"] / pk(k) for r, k in sel)      # GFLOP / (TFLOP/s) = ms
        return dict(gflop_per_step=fl, ms_per_step=ms, launches_per_step=n, tflops=fl / ms, frac=ideal_ms / ms, peak=fl / ideal_ms)
    out = dict(fwd_dgrad=group(("fwd", "dgrad")), with_wgrad_apportioned=group(("fwd", "dgrad", "wgrad")))
    for s in FFN_SITES:
        for k in ("fwd", "dgrad", "wgrad"):
            r = by.get("%s.%s" % (s, k))
            if r:
                out["%s.%s" % (s, k)] = dict(gflop=r["gflop_per_launch"], us=r["us_per_launch"], frac=r["tflops"] / pk(k), peak=pk(k))
    return out


def source_hash():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(site_labels, workload, dtype="bf16"):
    """Fabric-side bytes per launch (average over the launches of `site_labels`, e.g. 't.ffn_up.fwd') from the committed PMC
    passes of this same command (profiles/r*_pmc_sites*.json, made by tools/pmc_sites.py: separate --pmc FETCH_SIZE / WRITE_SIZE
    runs, FETCH_SIZE doubled per MI355X_MICROARCH.md, dispatches matched to the engine's launch log).  A table is used only if
    it is stamped with the hash of the kernel sources as they are now (``source_hash``) and with this workload: a measurement
    of other code is not reported.  None otherwise."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sites*.json"))):
        try:
            with open(path) as f:
                table = json.load(f)
        except (OSError, ValueError):
            continue
        if table.get("_source_hash") != source_hash() or table.get("_workload") != list(workload) or table.get("_dtype", "bf16") != dtype:
            continue
        if table.get("_mismatched_dispatches", 0):            # a pass that did not line up with the launch log: not a measurement
            continue
        cand = [table["sites"][k] for k in site_labels if k in table.get("sites", {}) and table["sites"][k].get("bytes_per_launch")]
        if cand:
            tot = sum(v["launches"] for v in cand)
            best = sum(v["bytes_per_launch"] * v["launches"] for v in cand) / max(tot, 1)
    return best


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(core, cfg, params, B, T, V, Fv, budget_s=20.0, max_threads=16):
    """Oracle forward+backward on the host cores, bounded to ~budget_s of CPU work: one warm-up and one
    probe at B=4 size the timed sample (B <= the benchmark batch) so that it fits the budget.
    Thread count is capped: PyTorch's CPU ops get slower, not faster, far beyond ~16 threads on these shapes."""
    from oracle import crct_oracle as O
    threads = max(1, min(os.cpu_count() or 1, max_threads))
    torch.set_num_threads(threads)
    sd = {k: p.detach().float().cpu().clone().requires_grad_(True) for k, p in core.named_parameters()}
    cpu_params = dict(params, device=torch.device("cpu"))
    small = S.make_batch(4, T, V, Fv, seed=99)
    O.oracle_step(sd, cfg, cpu_params, small)[0].backward()            # warm-up (page-in, thread pool)
    t0 = time.time()
    O.oracle_step(sd, cfg, cpu_params, small)[0].backward()
    probe = time.time() - t0
    Bs = int(max(4, min(B, 4 * budget_s / max(probe, 1e-3))))
    batch = S.make_batch(Bs, T, V, Fv, seed=1234)
    reps, dt = 0, 0.0
    while reps < 8 and (reps == 0 or dt < 0.5 * budget_s):          # fast hosts: repeat the sample up to ~budget_s / 2
        t0 = time.time()
        O.oracle_step(sd, cfg, cpu_params, batch)[0].backward()
        dt += time.time() - t0
        reps += 1
    return dict(value=Bs * reps / dt, unit="QA-pairs/s", cores=threads, kind="port",
                sample="%d x forward+backward of the fp32 CPU oracle (oracle/crct_oracle.py) at B=%d (sized from a B=4 probe of %.2f s "
                       "for a ~%.0f s budget), V=%d, T=%d, F_v=%d, dropout on; %.1f s in total on %d threads of %d logical cores (%s)"
                       % (reps, Bs, probe, budget_s, V, T, Fv, dt, threads, os.cpu_count() or 1, cpu_model()))


def launch_ranks(n, timeout_s=600.0):
    """``python bench.py --gpus N`` without a launcher: start one fresh child process per GPU -- what the reference does with
    ``mp.spawn(run_training_DDP, nprocs=num_proc)`` (CRCT/train.py:356-363) -- each running this same command line with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's JSON line, exit with the worst child's code.  The parent
    never touches a GPU and never exec()s: the children are ordinary subprocesses (a process that has initialised the GPU
    must not be replaced, and this one has not even done that).  ``timeout_s``: after that long the parent ends every rank it
    started (exactly those PIDs) and returns 124 -- ranks blocked in the RCCL bootstrap never return by themselves.  Every rank's
    stderr passes through the parent (line by line, as it comes), which keeps the last lines of each to say who was where."""
    import collections
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:                    # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    procs, tails, pumps = [], [], []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout carries the ONE JSON line; the other ranks' stdout goes to our stderr (they print nothing by design)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=subprocess.PIPE))
        tails.append(collections.deque(maxlen=12))

    def relay_out():                                 # rank 0's stdout, line by line as it comes (bytes: no decoding surprises)
        for line in procs[0].stdout:
            sys.stdout.buffer.write(line)
            sys.stdout.buffer.flush()

    def relay_err(r):
        for line in procs[r].stderr:
            tails[r].append(line)
            sys.stderr.buffer.write(line)
            sys.stderr.buffer.flush()
    pumps.append(threading.Thread(target=relay_out, daemon=True))
    pumps += [threading.Thread(target=relay_err, args=(r,), daemon=True) for r in range(n)]
    for t in pumps:
        t.start()
    worst, timed_out = 0, False
    t_end = time.monotonic() + timeout_s if timeout_s and timeout_s > 0 else None
    try:
        live = list(procs)
        while live:                                  # a rank that dies takes the job down: its peers would wait in a collective for ever
            time.sleep(0.05)
            if t_end is not None and time.monotonic() > t_end:
                timed_out = True
                break
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0:
                    if worst == 0:                   # the first failure is the job's exit code (the peers below are ended by us)
                        worst = rc
                    for q in live:
                        q.terminate()
    finally:
        for p in procs:                              # a rank that outlived a failed peer / the time limit: end exactly the PIDs started here
            if p.poll() is None:
                p.terminate()
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        for t in pumps:
            t.join(timeout=10)
    if timed_out or worst != 0:
        why = "no result after --rank-timeout-s %.0f s: every rank ended by the launcher" % timeout_s if timed_out else "a rank failed (exit code %d)" % worst
        sys.stderr.write("bench.py launcher: %s\n" % why)
        for r in range(n):
            rc = procs[r].poll()
            sys.stderr.write("  rank %d (exit %s), last stderr lines:\n" % (r, rc))
            for line in tails[r]:
                sys.stderr.write("    | " + line.decode(errors="replace").rstrip() + "\n")
        sys.stderr.flush()
    if timed_out:
        return 124
    return worst if 0 <= worst < 256 else 1


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us (the driver's 1-GPU command shape, `python bench.py --gpus N ...`): be the launcher.  Nothing
        # above this line touches a GPU (importing torch does not; the library is only dlopen()ed later, in the ranks).
        sys.exit(launch_ranks(a.gpus, a.rank_timeout_s))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world:
        raise SystemExit("bench.py --gpus %d does not match WORLD_SIZE=%d set by the launcher (run `python bench.py --gpus N`, which starts "
                         "its own ranks, or torch.distributed.run with --nproc-per-node N)" % (a.gpus, world))
    if a.launch_check:
        if os.environ.get("CRCT_LAUNCH_CHECK_FAIL_RANK") == str(rank):       # (only read under --launch-check: a rank that dies must take the job down)
            raise SystemExit(7)
        if os.environ.get("CRCT_LAUNCH_CHECK_HANG_RANK") == str(rank):       # (... and one that never comes back must run into --rank-timeout-s)
            sys.stderr.write("rank %d: waiting for a peer that never comes\n" % rank)
            sys.stderr.flush()
            time.sleep(3600)
        info = dict(rank=rank, local_rank=local, world=world, master="%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
                    ipc_legacy=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
        print(json.dumps(info) if rank == 0 else "rank %d: %s" % (rank, info), flush=True)
        return
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback for the measured path)")
    # nothing in the environment may change what the measured step launches: the library reads no CRCT_* variable any more, and
    # a stray one (an old A/B script) is an error rather than a silent no-op
    stray = sorted(k for k in os.environ if k.startswith("CRCT_") and k != "CRCT_BENCH_SHARE_GPU")
    if stray:
        raise SystemExit("bench.py: unset %s -- developer switches are command-line options (--site-policy ...), not environment variables" % ", ".join(stray))
    if os.environ.get("CRCT_BENCH_SHARE_GPU"):       # developer check of the N > 1 code path on a 1-GPU box (gloo, every rank on cuda:0)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_log = None
    if ((world > 1 or a.force_exchange) and rank == 0 and os.environ.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN")
            and "NCCL_DEBUG_FILE" not in os.environ and not os.environ.get("CRCT_BENCH_SHARE_GPU")):
        # RCCL has no getter for the channel count it chose: let rank 0 log its communicator INIT (only) into a file and read it
        # back for config.gradient_allreduce.rccl (crct/rccl.py: channels_from_debug_log)
        import tempfile
        rccl_log = os.path.join(tempfile.gettempdir(), "crct_rccl_init_%d.log" % os.getpid())
        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=rccl_log)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("CRCT_BENCH_SHARE_GPU"):
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    elif a.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)

    from crct.model import VisualDialogEncoder
    from crct.optim import get_optimizer, WarmupLinearScheduleNonZero
    from crct.step_adapter import forward as step_forward
    from crct.ddp import FlatGradDDP, AsyncStats
    from crct.input_pipeline import DevicePrefetcher

    cfg = CFG.vilbert_config(v_feature_size=a.feat)
    if a.no_dropout:
        cfg = CFG.vilbert_config(v_feature_size=a.feat, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                 v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
    params = CFG.default_params(device=dev, rank=rank, world_size=world, ddp=world > 1, batch_size=a.batch, seed=0, fp8=a.dtype == "fp8")
    if a.fp8_forward_only:
        params["fp8_backward"] = False
    if a.fp8_bf16_wgrad:
        params["fp8_wgrad"] = False
    if a.fp8_bf16_forward:
        params["fp8_forward"] = False
    if a.residual_bf16:
        params["residual_fp32"] = False
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.sync_stats = False                          # no .item() host syncs in the hot loop (train.py:178-183 does them)
    if a.no_dropout:
        core.cls_dropout = 0.0
    site_policy = []
    for item in filter(None, a.site_policy.split(",")):
        s, k, ph, c, sk = item.split(":")
        site_policy.append(dict(site=s, kind=k, phase=int(ph), cfg=int(c), split_k=int(sk)))
    core.site_policy = site_policy
    # with a gradient exchange the engine's weight gradients share ONE side stream: the exchange (auxiliary stream) then has a
    # hardware queue to itself -- a collective that really moves data must not sit in a compute stream's queue
    if a.wgrad_wgs >= 0:
        L.load().crct_gemm_group_max_workgroups(a.wgrad_wgs)
    if a.wgrad_target_wgs >= 0:
        core.wgrad_workgroups = (a.wgrad_target_wgs, a.wgrad_target_rows)
    if a.wgrad_concat >= 0:
        L.load().crct_gemm_group_concat(a.wgrad_concat)
    if a.wgrad_cfg >= 0:
        L.load().crct_gemm_group_wgrad_config(a.wgrad_cfg)
    if a.fp8_plain_mfma:
        L.load().crct_gemm_fp8_scaled_mfma(0)
    for item in filter(None, a.class_policy.split(",")):
        name, cfg_id = item.split("=")
        L.load().crct_gemm_class_config(L.CLASS_NAMES.index(name), int(cfg_id))
    if a.embed_scatter_split:
        L.load().crct_embed_scatter_split(1)
    wg_mode = a.wgrad_streams if a.wgrad_streams >= 0 else (2 if (world > 1 or a.force_exchange) else 1)
    core.stream_mode = (a.vis_stream, wg_mode)
    if a.wgrad_flush >= 0:
        core.wgrad_flush = a.wgrad_flush
    model.train()
    opt = get_optimizer(params, model)
    opt.overlap = not a.no_opt_overlap               # AdamW + gradient memset of step n overlap the forward of step n+1
    opt.fuse_zero_grad = bool(a.fuse_zero_grad)
    opt.lazy_zero_grad = not a.eager_zero_grad
    if a.adamw_wgs >= 0:
        opt.overlap_workgroups = a.adamw_wgs
    if a.adamw_groups >= 0:
        opt.launch_groups = a.adamw_groups
    if a.adamw_wide_first >= 0:
        opt.full_width_first = a.adamw_wide_first
    if a.opt_early and opt.overlap:
        opt.set_early(True)                          # ... or start per segment as soon as backward has finished it
    sched = WarmupLinearScheduleNonZero(opt, warmup_steps=params["warmup"], t_total=60000, min_lr=params["min_lr"])
    exchange = world > 1 or a.force_exchange
    ddp = None
    if exchange:                                     # attaches itself to the model
        ddp = FlatGradDDP(model, bucket_mb=a.bucket_mb, grad_dtype=torch.bfloat16 if a.grad_dtype == "bf16" else torch.float32)
        ddp.force_exchange = bool(a.force_exchange)
        if a.exchange_pack_all:
            ddp.direct_bf16_wgrad = False
        ddp.debug_skip = tuple(filter(None, a.exchange_skip.split(",")))
        if a.ghost_ranks > 1:
            if world != 1 or not a.force_exchange:
                raise SystemExit("--ghost-ranks needs --force-exchange on one GPU")
            ddp.ghost = dict(ranks=a.ghost_ranks, channels=a.ghost_channels, bus_GBps=a.ghost_bus_gbps)
    stats_red = AsyncStats(world, device=dev, ddp=ddp) if (exchange and "stats" not in a.exchange_skip) else None
    host_pool = [S.make_batch(a.batch, a.tokens, a.vis, a.feat, seed=1234 + rank + 97 * i) for i in range(8)]
    if a.emulate_ranks > 1:                          # what `emulate_ranks` data-parallel ranks see in one step, as ONE batch
        assert world == 1
        parts = [[S.make_batch(a.batch, a.tokens, a.vis, a.feat, seed=1234 + r + 97 * i) for r in range(a.emulate_ranks)] for i in range(8)]
        host_pool = [{k: torch.cat([p[k] for p in ps], 0) for k in ps[0]} for ps in parts]
    dev_pool = [stage(b, dev) for b in host_pool]
    host_feed = host_pool
    if a.host_feat == "bf16":                        # the data loader ships bf16 features: half the H2D bytes of the step
        host_feed = [dict(b, image_feat=b["image_feat"].to(torch.bfloat16)) for b in host_pool]

    def forever(pool):
        while True:
            for b in pool:
                yield b

    feeds = {"resident": forever(dev_pool), "sync": forever(host_pool)}
    prefetcher = None

    def feed_of(kind):
        nonlocal prefetcher
        if kind == "prefetch":
            if prefetcher is None:
                prefetcher = iter(DevicePrefetcher(forever(host_feed), dev, depth=2))
            return prefetcher
        return feeds[kind]

    cur = {"feed": feed_of(a.input)}

    def run_step():
        batch = next(cur["feed"])
        loss = step_forward(model, batch, params)[0]
        if stats_red is not None:
            # the reference's per-iteration stats exchange (train.py:181-189): copied and all-reduced on a side stream while
            # backward runs, collected at the end of the step (no collective between forward and backward on the main stream)
            stats_red.launch(core.last_stats[8:17])
        loss.backward()
        opt.step()
        opt.zero_grad()
        sched.step()
        if stats_red is not None:
            cur["stats"] = stats_red.result()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n):
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = run_step()
        cur["host_ms_per_step"] = (time.perf_counter() - t0) / n * 1e3       # host time to ENQUEUE a step (before the final sync)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        return dt, loss

    for _ in range(a.warmup):
        run_step()
    if a.launch_log:
        L.load().crct_launch_log_enable(1)
    dt, loss = timed(a.steps)
    if a.launch_log:
        lib = L.load()
        lib.crct_launch_log_enable(0)               # stop recording; the records stay readable
        recs = []
        for i in range(lib.crct_launch_log_count()):
            r = L.LaunchRec()
            lib.crct_launch_log_read(i, C.byref(r))
            recs.append(dict(site=L.SITE_NAMES[r.site] if r.site >= 0 else "group", kind=L.KIND_NAMES[r.kind], M=r.M, N=r.N, K=r.K, cfg=r.cfg,
                             split_k=r.split_k, n_problems=r.n_problems, flops=r.flops))
        with open(a.launch_log, "w") as f:
            json.dump(dict(steps=a.steps, source_hash=source_hash(), workload=[a.batch, a.vis, a.tokens, a.feat], dtype=a.dtype, launches=recs), f)
    host_ms = cur.get("host_ms_per_step")
    final_loss = float(loss.detach())
    # mean loss of the last timed step over the GLOBAL batch: the all-reduced stats when ranks exchange, else the local loss
    global_loss = float(cur["stats"][0]) if (stats_red is not None and cur.get("stats") is not None) else final_loss
    qa_per_s = a.batch * a.emulate_ranks * world * a.steps / dt

    sustained = None
    if a.sustained_s > 0:
        # the timed region above is K steps (0.15 s at the driver's K = 20): too short for any outside observer of GPU utilisation.
        # The same step, resident batches, for >= sustained_s seconds; reported next to the headline, never instead of it.
        n_sus = max(a.steps, int(a.sustained_s / max(dt / a.steps, 1e-4)) + 1)
        dt_s, _ = timed(n_sus)
        sustained = {"seconds": dt_s, "steps": n_sus, "ms_per_step": dt_s / n_sus * 1e3,
                     "qa_pairs_per_s": a.batch * a.emulate_ranks * world * n_sus / dt_s}
    h2d = None
    if a.input == "resident" and not a.no_h2d_leg:
        cur["feed"] = feed_of("prefetch")
        for _ in range(max(5, a.warmup // 2)):       # both slots' pinned / device buffers and the worker thread exist after two batches
            run_step()
        # K steps, twice; the lower is reported and both are listed: this leg shares the host with a packing thread, and a single
        # host hiccup (58 ms once in a round-4 collection: 10.5 instead of 7.7 ms per step) otherwise lands in a 0.15 s sample
        reps2 = [timed(a.steps)[0] for _ in range(2)]
        dt2 = min(reps2)
        feat_bytes = 2 if a.host_feat == "bf16" else 4
        h2d = {"input": "pageable host batches -> DevicePrefetcher (pinned staging, one async copy per batch on a copy stream, 2 slots), "
                        "image_feat shipped as %s" % a.host_feat,
               "ms_per_step": dt2 / a.steps * 1e3, "qa_pairs_per_s": a.batch * world * a.steps / dt2,
               "ms_per_step_of_each_repeat": [d / a.steps * 1e3 for d in reps2],
               "bytes_per_step_per_gpu": int(sum(v.numel() * (feat_bytes if k == "image_feat" else v.element_size())
                                                 for k, v in host_pool[0].items()))}
        cur["feed"] = feed_of("resident")
        prefetcher.close()                       # ends the worker thread (a suspended generator would keep it until exit)
        prefetcher = None

    comm = None
    if exchange:
        # evidence for the data-parallel exchange: the flat gradient buffer all-reduced on its own, after the
        # timed region (the step itself overlaps it with backward in >= bucket_mb pieces)
        used = int(max(hi for _, hi in core._engine.segments)) if core._engine is not None else core.flat_grads.numel()
        buf = core.flat_grads[:used]
        rc, aux = ddp.communicator(), core.aux_stream()

        def all_reduce():            # the step's own route: RCCL on the engine's auxiliary stream (torch.distributed only in the gloo developer mode)
            if rc is not None:
                rc.all_reduce_(buf, aux)
            else:
                dist.all_reduce(buf)
        torch.cuda.synchronize()
        for _ in range(2):
            all_reduce()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            all_reduce()
        torch.cuda.synchronize()
        ar = (time.perf_counter() - t1) / 5
        from crct import rccl as RC
        comm = {"rccl": {"version": RC.version()[1], "env": RC.env_seen(), "route": "direct ncclAllReduce (crct/rccl.py)" if rc is not None else "torch.distributed",
                         "direct_route_fallback_reason": getattr(ddp, "rccl_fallback", None),
                         "collectives_issued": getattr(rc, "collectives", None), "world": world,
                         "channels_logged_at_init": RC.channels_from_debug_log(rccl_log) if rccl_log else None,
                         "init_log_enabled_by_bench": bool(rccl_log)},      # NCCL_DEBUG=INFO / SUBSYS=INIT / DEBUG_FILE in env are then bench.py's
                "ghost": (dict(ddp.ghost, stand_in_ms_per_step=ddp.ghost_us / 1e3,
                               note="one-GPU prediction of the %d-rank step: per bucket a kernel of %d workgroups streams the payload twice through HBM and holds "
                                    "its stream for payload x 2 (N - 1) / N / bus bandwidth (crct_ghost_collective); nothing is reduced"
                                    % (ddp.ghost["ranks"], ddp.ghost["channels"])) if ddp.ghost else None),
                "allreduce_bytes": used * 4, "allreduce_ms": ar * 1e3,
                "bus_GBps": 2.0 * (world - 1) / world * used * 4 / ar / 1e9,
                "step_payload": {"dtype": a.grad_dtype, "bytes_per_step": used * (2 if a.grad_dtype == "bf16" else 4),
                                 "buckets": len(ddp._buckets or ()), "bucket_mb_of_fp32_gradients": a.bucket_mb,
                                 "collectives_issued_inside_the_backward_call": ddp.issued_inside_engine_call,
                                 "weight_gradients_written_as_bf16_by_the_gemms": bool(getattr(ddp, "packed_runs_only", False)),
                                 "route": "RCCL called directly on the engine's auxiliary stream" if rc is not None else "torch.distributed (%s)" % dist.get_backend(),
                                 "hardware_queue_classes_seen": getattr(core, "queue_classes", None), "weight_gradient_streams": wg_mode,
                                 "single_rank_forced": bool(a.force_exchange and world == 1)}}

    # profiled steps run on EVERY rank (they contain the collectives of a normal step); only rank 0 reads the stamps
    rows, sites = gemm_profile(run_step, a.profile_steps) if a.profile_steps > 0 else ([], [])
    # ... and so do the stamped steps behind config.critical_path (every library kernel with its own begin / end stamps)
    crit = critical_path(run_step, core, min(a.profile_steps, 2)) if a.profile_steps > 0 else None
    # the instruction each pass of the fp8 mode issues decides the peak its GEMMs are priced against: the block-scaled fp8 MFMA
    # (v_mfma_scale_f32_16x16x128_f8f6f4, unit scales) runs at twice the bf16 rate -- 5 PFLOP/s dense (MI355X_MICROARCH.md)
    kind_peak = {}
    if a.dtype == "fp8" and not a.fp8_plain_mfma:
        if not a.fp8_bf16_forward:
            kind_peak["fwd"] = PEAK_FP8_SCALED_TFLOPS
        if not a.fp8_forward_only:
            kind_peak["dgrad"] = PEAK_FP8_SCALED_TFLOPS
            if not a.fp8_bf16_wgrad:
                kind_peak["wgrad"] = PEAK_FP8_SCALED_TFLOPS
    if rank == 0:
        flop_qa = FLOP_PER_QA.get((a.vis, a.tokens, a.feat))
        out = {"metric": "QA-pairs/sec training step (whole node)", "value": qa_per_s, "unit": "QA-pairs/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "CRCT fwd+loss+bwd+AdamW, vilbert.json (v_feature_size=%d), batch %d/GPU, %d visual elems x %d-d, "
                                      "%d text tokens, dropout 0.1, L1 regression loss%s" % (a.feat, a.batch, a.vis, a.feat, a.tokens,
                                      ("; bf16 forward and weight gradients, fp8 data gradients (e5m2 x e4m3) of every encoder Linear" if (a.fp8_bf16_forward and a.fp8_bf16_wgrad) else
                                       "; bf16 forward, fp8 backward: data gradient e5m2 x e4m3, weight gradient e5m2 x e4m3 of every encoder Linear" if a.fp8_bf16_forward else
                                       "; fp8 (e4m3) forward GEMMs of every encoder Linear, bf16 backward" if a.fp8_forward_only else
                                       "; every encoder Linear in fp8: forward e4m3 x e4m3, data gradient e5m2 x e4m3" +
                                       (", bf16 weight gradients" if a.fp8_bf16_wgrad else ", weight gradient e5m2 x e4m3"))
                                      if a.dtype == "fp8" else "") + ("; residual stream stored as bf16 (params['residual_fp32'] = False)" if a.residual_bf16 else ""),
                          "residual_stream": "bf16" if a.residual_bf16 else "fp32",
                          "global_batch": a.batch * a.emulate_ranks * world, "parallelism": "dp%d" % world, "final_loss": final_loss,
                          "global_loss": global_loss, "input": a.input, "host_enqueue_ms_per_step": host_ms,
                          "sustained": sustained, "h2d_inclusive": h2d, "gradient_allreduce": comm, "site_policy": a.site_policy or None, "class_policy": a.class_policy or None,
                          "critical_path": crit,
                          # every developer switch that changes what the step launches (all defaults: the measured step is the product)
                          "schedule": {"weight_gradient_streams": wg_mode, "visual_stream": a.vis_stream, "wgrad_flush": a.wgrad_flush if a.wgrad_flush >= 0 else "default (1)",
                                       "wgrad_cfg": a.wgrad_cfg if a.wgrad_cfg >= 0 else "default (4)", "wgrad_max_workgroups": a.wgrad_wgs if a.wgrad_wgs >= 0 else None,
                                       "wgrad_target_workgroups": [a.wgrad_target_wgs, a.wgrad_target_rows] if a.wgrad_target_wgs >= 0 else "default (96, 3000)",
                                       "wgrad_concat": a.wgrad_concat if a.wgrad_concat >= 0 else 0, "optimizer_overlap": not a.no_opt_overlap, "optimizer_early": bool(a.opt_early),
                                       "adamw_workgroups": a.adamw_wgs if a.adamw_wgs >= 0 else "default (256)", "eager_zero_grad": bool(a.eager_zero_grad),
                                       "fuse_zero_grad": bool(a.fuse_zero_grad), "exchange_skip": a.exchange_skip or None},
                          "gemm_sites": sites, "gemm_variants": rows}}
        if h2d is not None:      # the PCIe-inclusive rate next to `value` (SURVEY.md 8d defines the step with its H2D copy; the bench contract's `value` is HBM-resident)
            out["value_h2d_inclusive"] = h2d["qa_pairs_per_s"]
            out["ms_per_step_h2d_inclusive"] = h2d["ms_per_step"]
        if flop_qa:
            out["config"]["step_model_flops_frac_of_bf16_peak"] = qa_per_s * flop_qa / (world * PEAK_BF16_TFLOPS * 1e12)
        if sites:
            ffn = ffn_roofline(sites, a.profile_steps, kind_peak)
            grp = ffn["fwd_dgrad"]
            dom = max(rows, key=lambda r: r["ms_per_step"])
            # the committed PMC tables are of the default step (fp32 residual stream): no traffic figure for the bf16-residual variant
            traffic = None if a.residual_bf16 else pmc_traffic(["%s.%s" % (s, k) for s in FFN_SITES for k in ("fwd", "dgrad")], (a.batch, a.vis, a.tokens, a.feat), a.dtype)
            out["roofline"] = {"bound": "mfma", "kernel": "FFN GEMMs: text / visual FFN-up + FFN-down, forward + data gradient (%d launches per step)"
                                                             % round(grp["launches_per_step"]),
                               "achieved": grp["tflops"], "peak": grp["peak"], "unit": "TFLOP/s", "frac": grp["frac"], "traffic": traffic,
                               "peak_by_pass": {k: kind_peak.get(k, PEAK_BF16_TFLOPS) for k in ("fwd", "dgrad", "wgrad")},
                               "gflop_per_launch": grp["gflop_per_step"] / grp["launches_per_step"],
                               "us_per_launch": grp["ms_per_step"] * 1e3 / grp["launches_per_step"], "ffn": ffn,
                               "largest_kernel_class": {"kernel": dom["kernel"], "tflops": dom["tflops"], "frac": dom["tflops"] / PEAK_BF16_TFLOPS,
                                                        "ms_per_step": dom["ms_per_step"]},
                               "timing": "kernel begin / end stamps (hipExtLaunchKernelGGL start / stop events), %d profiled steps, in the step "
                                         "(other streams' kernels run beside each launch)" % a.profile_steps}
        if not a.no_cpu_baseline:      # rank 0 only, after the timed region (its peers wait at the closing barrier meanwhile)
            out["cpu_baseline"] = cpu_baseline(core, cfg, params, a.cpu_batch, a.tokens, a.vis, a.feat)
        print(json.dumps(out), flush=True)
    if prefetcher is not None:
        prefetcher.close()
    if world > 1:
        dist.barrier()
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
