#!/usr/bin/env python3
"""Headline benchmark: concurrent 48 kS/s DMR channels sustained end-to-end on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (one kernel: every channel's wavefront filters + slices its samples
and then runs the DMR decoder over the symbols it produced; --split-stages launches the two stages separately)
over one batch of synthetic input that is already resident in HBM: by default BASELINE.json
configs[2], 16 384 DMR channels x 3.96 s of 48 kS/s audio per GPU, full chain incl. BPTC(196,96).
State (filter history, timing recovery, decoder phase) carries from step to step exactly as in a
continuous stream; the input buffer is periodic so the stream is seamless.

One process per GPU.  Started as plain `python bench.py --gpus N` with N > 1 it re-executes itself under
`torch.distributed.run` with N ranks (and refuses when the box has fewer GPUs); a line whose n_gpus differs from
--gpus is never printed.  Channels shard by index, no collective on the data path: RCCL only carries the barrier
and the MAX / SUM of the report.  `--scaling weak` (default): --channels per GPU; `--scaling strong`:
--total-channels split over the ranks (BASELINE configs[4]: --workload mixed --scaling strong --total-channels 65536).

One JSON line on rank 0: value = whole-job real-time 48 kS/s channels = samples/s / 48 000, plus
`roofline` for the dominant kernel (timed with HIP events on its own stream inside the timed region),
`cpu_baseline` (the oracle's scalar restatement of the reference pipe on this box's host cores, bounded sample,
rank 0 at N = 1 only) and `other_configs`: the other single-GPU BASELINE configs timed on the same lease
(N = 1 only, after the headline's timed region; --no-other-configs skips them).
"""
import argparse
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable with a copy kernel
SAMPLE_RATE = 48000

WORKLOADS = {
    # name: (protocol of the synthetic signal, engine kwargs, description)
    "dmr_full": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="dmr"),
                 "full chain rrc(wide)->gfsk(10)->dmr_decoder incl. BPTC(196,96) (BASELINE configs[2])"),
    "ysf_full": ("ysf", dict(rrc="wide", demod="gfsk", sps=10, proto="ysf"),
                 "full chain rrc(wide)->gfsk(10)->ysf_decoder incl. Golay/Viterbi (BASELINE configs[3])"),
    "rrc_gfsk": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="none", keep_filtered=True),
                 "rrc(wide) materialised + gfsk(10), float path (BASELINE configs[1])"),
    "rrc_gfsk_fast": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="none", keep_filtered=True, fast_fir=True),
                      "rrc(wide) materialised with the FMA FIR (1e-6 float tolerance of configs[1]) + gfsk(10)"),
    "rrc_gfsk_one": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="none", keep_filtered=True, one_launch=True),
                     "rrc(wide) + gfsk(10) in ONE launch (DH_FLAG_ONE_LAUNCH): dibits bit-exact, filtered samples from the split-f16 matrix-core FIR -- "
                     "2.5e-6 of max(|ref|, rms), NOT configs[1]'s 1e-6: shown for what one kernel reaches, not as the configs[1] number"),
    "rrc_gfsk_fast_one": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="none", keep_filtered=True, one_launch=True, fast_fir=True),
                          "rrc(wide) + gfsk(10) in ONE launch with the f32 FMA chain on the matrix cores (DH_FLAG_ONE_LAUNCH | DH_FLAG_FAST_FIR): floats within 1e-6 "
                          "AND dibits bit-exact -- BASELINE configs[1] in one kernel"),
    "dmr_fast": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="dmr", fast_fir=True),
                 "full DMR chain with the FMA FIR (float outputs 1e-6, dibits not guaranteed bit-exact)"),
    # narrow RRC -> gfsk -s 20 -> nxdn_decoder (examples/nxdn48-decoder.sh)
    "nxdn_full": ("nxdn", dict(rrc="narrow", demod="gfsk", sps=20, proto="nxdn"),
                  "full chain rrc(narrow)->gfsk(20)->nxdn_decoder (NXDN48)"),
    # examples/dstar-decoder.sh: fsk_demodulator -s 10 | dstar_decoder (no RRC stage on this path)
    "dstar_full": ("dstar", dict(rrc="none", demod="fsk", sps=10, proto="dstar"),
                   "full chain fsk(10)->dstar_decoder (D-Star)"),
    # examples/pocsag-decoder.sh: fsk_demodulator -i -s 40 | pocsag_decoder
    "pocsag_full": ("pocsag", dict(rrc="none", demod="fsk", sps=40, proto="pocsag", invert=True),
                    "full chain fsk(40, inverted)->pocsag_decoder (POCSAG 1200)"),
    # the receiver front-end in front of the chain -- int16 I / Q in, polar discriminator + DC
    # blocker (dh_frontend_s16, own specification) as a pre-stage kernel, then the headline chain
    "dmr_iq_full": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="dmr"),
                    "int16 I/Q -> front-end (FM discriminator + DC block) -> rrc(wide)->gfsk(10)->dmr_decoder"),
    # BASELINE configs[4]: half the channels DMR, half YSF, one engine (and one launch per push) each
    "mixed": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="dmr"),
              "half DMR + half YSF channels, full chains (BASELINE configs[4])"),
}
UNITS = {"dmr": 132, "ysf": 40, "nxdn": 50, "dstar": 198, "pocsag": 148}     # bursts / frames per step: ~3.96-4 s


def _profile_config(pdir, name, workload):
    """(channels, samples per channel) the PMC passes of profiles/<name> ran on: the `## config:` line tools/profile_gpu.sh
    writes into the summary, else the bench line captured under the tracer beside it (<prefix>_bench_<workload>_under_trace.json)."""
    for line in open(os.path.join(pdir, name)):
        if line.startswith("## config:"):
            kv = dict(t.split("=", 1) for t in line.split()[2:] if "=" in t)
            return int(kv["channels_per_gpu"]), int(kv["samples_per_channel"])
    beside = os.path.join(pdir, name.replace("_%s_pmc.txt" % workload, "_bench_%s_under_trace.json" % workload))
    try:
        cfg = json.loads(open(beside).read().strip().splitlines()[-1])["config"]
        return int(cfg["channels_per_gpu"]), int(cfg["samples_per_channel_per_step"])
    except (OSError, ValueError, KeyError, IndexError):
        return None


def profiled_counters(workload, channels, T, part0=False):
    """Counters of the dominant kernel from the committed rocprofv3 PMC passes (bench.py cannot collect counters itself):
    HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KiB, per-dispatch average; MI355X_MICROARCH.md), and the share of
    the SIMDs' cycles on which a vector instruction issued.  The passes name the configuration they ran on (_profile_config);
    per-launch byte and instruction counts are proportional to channels x samples and are SCALED to this run's -- a summary
    that does not say what it ran on is not used.  Newest round first.
    part0: the push goes out as two launches; take the lines of the first one (kernel template argument PART = 0)."""
    pdir = os.path.join(ROOT, "profiles")
    kern = "k_rrc_demod" if workload in ("rrc_gfsk_one", "rrc_gfsk_fast_one") else "k_rrc_tile" if workload.startswith("rrc_gfsk") else "k_chain"
    want = lambda line: kern in line and (not part0 or ", 10, 0>" in line)
    names = sorted((f for f in os.listdir(pdir) if f.endswith("_%s_pmc.txt" % workload) or (workload == "dmr_full" and f.endswith("_chain_pmc.txt"))), reverse=True)
    for name in names:
        ran = _profile_config(pdir, name, workload)
        if not ran:
            continue
        scale = (channels * float(T)) / (ran[0] * float(ran[1]))
        c = {}
        pass_ms = None                           # avg_launch_ms the pass's own bench line reported (tools/profile_gpu.sh, round 6 on)
        for line in open(os.path.join(pdir, name)):
            if line.startswith("## config:"):
                kv = dict(t.split("=", 1) for t in line.split()[2:] if "=" in t)
                try:
                    pass_ms = float(kv["avg_launch_ms"])
                except (KeyError, ValueError):
                    pass_ms = None
            if want(line) and "avg=" in line:
                if " GRBM_GUI_ACTIVE " in line and "GRBM_GUI_ACTIVE" not in c and pass_ms:
                    c["_grbm_pass_ms"] = pass_ms
                for key in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE",
                            "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
                    if " %s " % key in line and key not in c:
                        c[key] = float(line.split("avg=")[1].split()[0])
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out = {"traffic": (c["FETCH_SIZE"] * 2.0 * 1024.0 + c["WRITE_SIZE"] * 1024.0) * scale,
                   "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of this workload at %d channels x %d samples%s)"
                                     % (name, ran[0], ran[1], "" if scale == 1.0 else ", scaled x%.4g to this run's size" % scale)}
            if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                out["valu_issue_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
                out["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                      "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
                if k in c:
                    out[k.lower() + "_per_launch"] = c[k] * scale
            if "GRBM_GUI_ACTIVE" in c:
                out["gpu_cycles_per_launch_profiled"] = c["GRBM_GUI_ACTIVE"] / 8.0 * scale      # (summed over the 8 XCDs)
                out["profile_scale"] = scale
                if "_grbm_pass_ms" in c:
                    out["profiled_avg_launch_ms"] = c["_grbm_pass_ms"]
            out["pmc_source"] = "profiles/" + name
            return out
    return {}


_COPY_GBS = {}


def copy_ceiling(torch, device, ctx):
    """What a plain streaming kernel reaches on this lease: the library's own copy kernel (dh_debug_copy: 16 bytes per lane,
    non-temporal loads and stores) over 2 GiB, read + write bytes over its duration, torch's copy_ on the same buffers, and the
    library's READ-ONLY stream (dh_debug_copy with a null destination; the chain kernels read 93 % of their HBM bytes) -- the
    achievable HBM ceiling SURVEY.md section 8(d) asks for beside the 8 TB/s of the data sheet.  The largest of the three is the
    ceiling (a ceiling below what some kernel reaches is not one)."""
    key = str(device)
    if key not in _COPY_GBS:
        n = 1 << 29                                  # floats: 2 GiB in, 2 GiB out
        a = torch.empty(n, dtype=torch.float32, device=device).normal_()
        b = torch.empty_like(a)
        stream = ctx.mem.stream()

        def own():
            assert ctx.lib.dh_debug_copy(ctx.mem.ptr(a), ctx.mem.ptr(b), n * 4, stream) == 0
        def own_read():
            assert ctx.lib.dh_debug_copy(ctx.mem.ptr(a), None, n * 4, stream) == 0
        rates = {}
        for name, fn in (("dh_debug_copy", own), ("torch.Tensor.copy_", lambda: b.copy_(a)), ("dh_debug_copy read-only", own_read)):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            e1.synchronize()
            rates[name] = 5 * (1.0 if name.endswith("read-only") else 2.0) * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        assert bool((a[:: 4097] == b[:: 4097]).all())
        _COPY_GBS[key] = (max(rates.values()), rates)
        del a, b
        torch.cuda.empty_cache()
    return _COPY_GBS[key]


def oracle_kw(proto):
    if proto == "dstar":
        return dict(proto=5, rrc=0, levels=2, sps=10)
    if proto == "pocsag":
        return dict(proto=4, rrc=0, levels=2, sps=40, invert=True)
    return dict(proto={"dmr": 1, "ysf": 2, "nxdn": 3}[proto], **(dict(rrc=2, sps=20) if proto == "nxdn" else {}))


def effective_cores():
    """(visible, effective): os.cpu_count() vs what this process may actually use = min(affinity mask, cgroup CPU quota)."""
    visible = os.cpu_count() or 1
    eff = visible
    try:
        eff = min(eff, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                eff = min(eff, max(1, int(math.ceil(float(quota) / period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return visible, max(1, eff)


def outputs_sha(syms, sym_count, frames, frame_count):
    """SHA-256 over every channel's dibits and decoder bytes of one push (counts included)."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(sym_count, dtype=np.uint32).tobytes())
    h.update(np.ascontiguousarray(frame_count, dtype=np.uint32).tobytes())
    for b in range(len(sym_count)):
        h.update(np.ascontiguousarray(syms[b, :sym_count[b]]).tobytes())
        h.update(np.ascontiguousarray(frames[b, :frame_count[b]]).tobytes())
    return h.hexdigest()


def cpu_baseline(x_host_fn, proto, budget_s=12.0):
    """Time the oracle (unpinned scalar CPU restatement of the reference pipe) on the host cores this process may use."""
    from oracle import oracle as O
    visible, cores = effective_cores()
    probe = x_host_fn(1)
    n = probe.shape[1]
    O.chain(probe[:, : min(n, 24000)], threads=1, **oracle_kw(proto))                     # page in
    t0 = time.perf_counter()
    O.chain(probe[:, : min(n, 96000)], threads=1, **oracle_kw(proto))
    per_sample = (time.perf_counter() - t0) / min(n, 96000)
    chans = max(cores, int(budget_s / (per_sample * n) * cores))                          # ~budget_s with every thread busy
    chans = min(chans, 16384)
    x = x_host_fn(chans)
    t0 = time.perf_counter()
    ref = O.chain(x, threads=cores, **oracle_kw(proto))
    dt = time.perf_counter() - t0
    ref_sha = outputs_sha(ref["syms"], ref["sym_count"], ref["out"], ref["out_count"])
    rate = x.size / dt
    per_core, single = rate / 1e6 / cores, 1e-6 / per_sample
    return {"value": rate / SAMPLE_RATE, "unit": "channels", "msamples_per_s": rate / 1e6,
            "msamples_per_s_per_core": per_core, "single_thread_msamples_per_s": single,
            "cores": cores, "cores_visible": visible, "cores_effective": cores, "oversubscribed": bool(per_core < 0.5 * single),
            "kind": "port", "outputs_sha256": ref_sha, "channels_hashed": int(chans),
            "what": "CPU restatement (unpinned port): oracle/ C code, not the reference binaries -- rrc_filter / gfsk_demodulator / "
                    "*_decoder need csdr, which this image lacks, so examples/dmr-decoder.sh itself cannot be timed here",
            "sample": "%d channels x %d samples of the same synthetic workload through oracle/ (scalar C restatement of "
                      "[rrc_filter|]g/fsk_demodulator|%s_decoder, bit-exact with the GPU path), %d pthreads, %.1f s wall"
                      % (chans, x.shape[1], proto, cores, dt)}


def reference_fec_baseline(torch, ctx, seconds=1.5):
    """The one piece of the reference that builds in this image -- its FEC sources, compiled in place into
    oracle/_ref/libdigiham_ref_fec.so -- timed single-threaded on this box beside the product's batch entries on the same
    blocks (inputs resident in HBM, outputs compared).  kind = "reference"."""
    import numpy as np
    from oracle import oracle as O
    if O.ref() is None:
        return {"error": "oracle/_ref/libdigiham_ref_fec.so not built on this box"}
    rng = np.random.default_rng(5)
    out = {"kind": "reference", "cores": 1, "what": "src/dmr_decoder/bptc_196_96.c and src/ysf_decoder/trellis.c of the reference, compiled in place "
                                                    "(oracle/Makefile), one thread; gpu_* = dh_bptc_196_96 / dh_trellis on the same blocks"}
    n = 200000
    pay = rng.integers(0, 256, (n, 25)).astype(np.uint8)
    pay[:, 24] &= 0xF0
    t0 = time.perf_counter(); r_out, r_ok = O.bptc_196_96(pay[:20000], which="ref"); per = (time.perf_counter() - t0) / 20000
    k = int(min(n, max(20000, seconds / per)))
    t0 = time.perf_counter(); r_out, r_ok = O.bptc_196_96(pay[:k], which="ref"); dt = time.perf_counter() - t0
    d_in = ctx.mem.from_numpy(pay[:k]); d_out = ctx.mem.zeros((k, 12), np.uint8); d_ok = ctx.mem.zeros((k,), np.uint8)

    def launch():
        rc = ctx.lib.dh_bptc_196_96(ctx.mem.ptr(d_in), ctx.mem.ptr(d_out), ctx.mem.ptr(d_ok), k, ctx.mem.stream())
        assert rc == 0
    launch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        launch()
    torch.cuda.synchronize(); gdt = (time.perf_counter() - t0) / 20
    g_out, g_ok = ctx.mem.to_numpy(d_out), ctx.mem.to_numpy(d_ok)
    good = r_ok.astype(bool)
    out["bptc_196_96"] = {"blocks": k, "reference_blocks_per_s": k / dt, "gpu_blocks_per_s": k / gdt,
                          "identical": bool((g_ok == r_ok).all() and (g_out[good] == r_out[good]).all())}
    m = 100000
    packed = rng.integers(0, 256, (m, 45)).astype(np.uint8)
    t0 = time.perf_counter(); t_out, t_metric = O.trellis(packed[:10000], 180, which="ref"); per = (time.perf_counter() - t0) / 10000
    k = int(min(m, max(10000, seconds / per)))
    t0 = time.perf_counter(); t_out, t_metric = O.trellis(packed[:k], 180, which="ref"); dt = time.perf_counter() - t0
    ob = (180 + 7) // 8
    d_in = ctx.mem.from_numpy(packed[:k]); d_out = ctx.mem.zeros((k, ob), np.uint8); d_m = ctx.mem.zeros((k,), np.uint8)

    def launch2():
        rc = ctx.lib.dh_trellis(ctx.mem.ptr(d_in), 45, 180, ctx.mem.ptr(d_out), ob, ctx.mem.ptr(d_m), k, ctx.mem.stream())
        assert rc == 0
    launch2(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        launch2()
    torch.cuda.synchronize(); gdt = (time.perf_counter() - t0) / 20
    out["trellis_180"] = {"blocks": k, "reference_blocks_per_s": k / dt, "gpu_blocks_per_s": k / gdt,
                          "identical": bool((ctx.mem.to_numpy(d_out) == t_out).all() and (ctx.mem.to_numpy(d_m) == t_metric).all())}
    return out


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: become N ranks, or fail loudly."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        sys.exit("bench.py: --gpus %d requested but this box has %d GPU(s); not reporting a line for fewer GPUs" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


class Job:
    """The engines of one workload on this rank's GPU, with their inputs resident in HBM."""

    def __init__(self, torch, ctx, device, workload, channels, rank, split_stages=False, streams=1, units=0, overlap=False):
        from digiham_amd import api, synth_torch
        self.torch, self.workload = torch, workload
        proto, kw, self.desc = WORKLOADS[workload]
        if split_stages:
            kw = dict(kw, split_stages=True)
        if overlap:
            kw = dict(kw, overlap_pushes=True)
        self.kw, self.proto = kw, proto
        parts = [(proto, kw, channels)] if workload != "mixed" else \
                [("dmr", kw, channels[0]), ("ysf", dict(kw, proto="ysf"), channels[1])]
        self.parts = []
        for i, (p, k, B) in enumerate(parts):
            if B == 0:
                continue
            x, info = synth_torch.make_batch(torch, device, p, B, units or UNITS[p], seed=1000 * (i + 1) + 7919 * rank, sps=k["sps"])
            T = info["samples_per_channel"]
            iq_pipe = workload.endswith("_iq_full") and streams > 1      # int16 I/Q: the front-end of step k + 1 on its own stream beside the chain kernel of step k
            stream = torch.cuda.Stream(device) if streams > 1 and not iq_pipe else None
            if stream is not None:
                with torch.cuda.stream(stream):
                    eng = api.Engine(B, T, ctx=ctx, **k)
            else:
                eng = api.Engine(B, T, ctx=ctx, **k)
            part = {"proto": p, "kw": k, "B": B, "T": T, "x": x, "eng": eng, "stream": stream}
            if workload.endswith("_iq_full"):                       # the same audio as FM on a carrier, int16 I / Q, resident in HBM
                iq = torch.empty((B, 2 * T), dtype=torch.int16, device=device)
                for c0 in range(0, B, 1024):
                    ph = torch.cumsum(x[c0:c0 + 1024].double() * (math.pi * 0.35) + math.pi * 0.01, dim=1)
                    iq[c0:c0 + 1024, 0::2] = (12000.0 * torch.cos(ph)).round().to(torch.int16)
                    iq[c0:c0 + 1024, 1::2] = (12000.0 * torch.sin(ph)).round().to(torch.int16)
                    del ph
                part["iq"] = iq
                part["fe_state"] = torch.zeros((B, 4), dtype=torch.float32, device=device)
                part["x"] = torch.empty_like(x)                     # the front-end's output buffer = the engine's input
                part["ctx"] = ctx
                if iq_pipe:
                    # two float buffers: the front-end writes one (its own stream) while the chain kernel reads the other; events order
                    # "converted before pushed" and "pushed before overwritten" -- what a receiver with a ring of buffers does anyway
                    part["xbuf"] = [part["x"], torch.empty_like(x)]
                    part["fe_stream"] = torch.cuda.Stream(device)
                    part["fe_done"] = [torch.cuda.Event(), torch.cuda.Event()]
                    part["push_done"] = [torch.cuda.Event(), torch.cuda.Event()]
                    part["k"] = 0
                    part["fe_stream"].wait_stream(torch.cuda.current_stream(device))      # (the I/Q samples and the front-end's state were written on this stream)
            self.parts.append(part)
        self.samples_per_step = float(sum(p["B"] * p["T"] for p in self.parts))

    def step(self):
        for p in self.parts:
            if "fe_stream" in p:
                torch = self.torch
                c, mem = p["ctx"], p["ctx"].mem
                b = p["k"] & 1
                p["k"] += 1
                xb = p["xbuf"][b]
                cur = torch.cuda.current_stream(p["iq"].device)
                with torch.cuda.stream(p["fe_stream"]):
                    p["fe_stream"].wait_event(p["push_done"][b])            # the push that last read this buffer (two steps ago) is through
                    rc = c.lib.dh_frontend_s16(mem.ptr(p["iq"]), 2 * p["T"], mem.ptr(xb), p["T"], mem.ptr(p["fe_state"]), p["B"], p["T"], 2, 1, mem.stream())
                    assert rc == 0, "dh_frontend_s16 failed"
                    p["fe_done"][b].record(p["fe_stream"])
                cur.wait_event(p["fe_done"][b])
                p["eng"].push(xb)
                p["push_done"][b].record(cur)
                continue
            if "iq" in p:
                c, mem = p["ctx"], p["ctx"].mem
                rc = c.lib.dh_frontend_s16(mem.ptr(p["iq"]), 2 * p["T"], mem.ptr(p["x"]), p["T"], mem.ptr(p["fe_state"]), p["B"], p["T"], 2, 1, mem.stream())
                assert rc == 0, "dh_frontend_s16 failed"
            p["eng"].push(p["x"])

    def sync(self):
        for p in self.parts:
            p["eng"].sync()                 # raises on any output-buffer overflow

    def close(self):
        for p in self.parts:
            p["eng"].close()

    def timed(self, steps, warmup, barrier=lambda: None):
        """W untimed steps, then exactly `steps` steps between barrier + synchronize on both sides.  Returns seconds."""
        torch = self.torch
        for p in self.parts:
            p["eng"].timing_enable(max(steps, 1))
        for _ in range(warmup):
            self.step()
        self.sync()
        for p in self.parts:
            p["eng"].timing_read()          # drop warm-up timings
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        self.sync()
        return dt

    def step_algorithmic_bytes(self):
        """SURVEY.md section 8(d) for the WHOLE step, every engine of it: 4 B in per sample (f32 audio, or an int16 I/Q pair) +
        1 B per symbol + the decoders' output of the last step (+ 4 B per sample where the filtered signal is materialised)."""
        total = 0.0
        for p in self.parts:
            kw = p["kw"]
            total += p["B"] * p["T"] * (4.0 + (4.0 if kw.get("keep_filtered") else 0.0) + 1.0 / kw["sps"])
            if kw["proto"] != "none":
                total += int(p["eng"].frames()[1].sum())
        return total

    def roofline(self, split_stages=False, step_ms=None):
        """Dominant kernel of the first part: its algorithmic bytes per launch over its average
        launch duration, from the HIP events the engine records on its own stream around every launch."""
        import numpy as np
        p = self.parts[0]
        kw, B, T = p["kw"], p["B"], p["T"]
        first_ms, first_ch = p["eng"].timing_read_split()     # pushes that went out as two launches (large DMR / YSF engines)
        rrc_ms, slicer_ms, dec_ms = p["eng"].timing_read()
        frame_bytes = int(p["eng"].frames()[1].sum()) if kw["proto"] != "none" else 0      # decoder output of the last step
        alg_bytes = B * T * 4.0 + B * (T / float(kw["sps"]))          # input f32 (4 B/sample) + dibits out (1 B per sps samples)
        if kw.get("keep_filtered") and kw.get("one_launch"):
            # DH_FLAG_ONE_LAUNCH (engine_impl.hpp, fused_keep): the error-bounded slicer kernel also stores the filtered samples
            dom_ms = float(np.mean(slicer_ms)) if len(slicer_ms) else float("nan")
            alg_bytes = B * T * 8.0 + B * (T / float(kw["sps"]))                          # 4 B in + 4 B out per sample + 1 B per symbol (8.1)
            dom_name = "k_rrc_demod"
        elif kw.get("keep_filtered"):
            dom_ms = float(np.mean(rrc_ms)) if len(rrc_ms) else float("nan")              # unfused config: the RRC kernel dominates
            alg_bytes = B * T * 8.0                                                       # 4 B in + 4 B out per sample
            dom_name = "k_rrc_tile"
        else:
            dom_ms = float(np.mean(slicer_ms)) if len(slicer_ms) else float("nan")
            # one launch for slicer + decoder exists for these chains (engine.hip: launch_chain)
            chained = ((kw["proto"] in ("dmr", "ysf") or (kw["proto"] == "dstar" and kw["rrc"] == "none")) and kw["sps"] == 10
                       or (kw["proto"] == "nxdn" and kw["rrc"] == "narrow")) and not split_stages
            dom_name = "k_chain" if chained else "k_rrc_demod"
            if chained:
                alg_bytes += frame_bytes            # + decoder output (<= 27 B per 1440 samples for DMR)
        group = None
        if dom_name == "k_chain" and len(first_ch) and first_ch.min() > 0:
            # the push is two launches of the same kernel (template argument PART 0 / 1, listed apart by rocprofv3): the
            # dominant one is PART 0 on the engine's high-priority stream; bytes and duration are ITS share and ITS events
            share = float(first_ch[0]) / B
            group = {"launches_per_push": 2, "first_launch_channels": int(first_ch[0]),
                     "note": "overlapped pushes (DH_FLAG_OVERLAP_PUSHES): the two launches of a push and those of its neighbours "
                             "share the chip, so a launch's own duration includes time it spends beside the others",
                     "whole_push_algorithmic_bytes": alg_bytes, "step_period_ms": step_ms,
                     "whole_push_frac": (alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_ms else None}
            alg_bytes *= share
            dom_ms = float(np.mean(first_ms))
            B = int(first_ch[0])
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        taps = {"wide": 81, "narrow": 161, "none": 0}[kw["rrc"]]
        fir_flops = B * T * 2.0 * taps              # one mul + one add per tap and sample, unfused
        pc = profiled_counters(self.workload, p["B"], T, part0=group is not None)
        if len(self.parts) > 1:
            # two engines (mixed): the step moves both parts' bytes -- the sum of the parts' profiled traffic, each scaled to its size
            pcs = [profiled_counters("%s_full" % q["kw"]["proto"], q["B"], q["T"]) for q in self.parts]
            if all("traffic" in q for q in pcs):
                pc = dict(pc)
                pc["traffic"] = sum(q["traffic"] for q in pcs)
                pc["traffic_source"] = "WHOLE STEP, sum over its engines (compare with algorithmic_bytes_per_step): " + "; ".join(q["traffic_source"] for q in pcs)
                # algorithmic bytes of the whole step: 4 B in + 1 B per sps samples per channel, + what the decoders put out
                pc["algorithmic_bytes_per_step"] = sum(q["B"] * q["T"] * 4.0 + q["B"] * (q["T"] / float(q["kw"]["sps"])) +
                                                       (int(q["eng"].frames()[1].sum()) if q["kw"]["proto"] != "none" else 0) for q in self.parts)
        if group is not None and "traffic" in pc:
            # the committed passes profiled whole pushes (one launch of all channels); PART 0 of an overlapped push takes `share` of them
            pc = dict(pc)
            pc["traffic"] *= share
            pc["traffic_source"] += "; x%.3f: the first launch's share of the channels" % share
            for k in list(pc):
                if k.endswith("_per_launch") or k in ("gpu_cycles_per_launch_profiled", "profile_scale"):
                    pc[k] *= share
        mean = lambda a: float(np.mean(a)) if len(a) else None
        f16 = (dom_name == "k_chain" or kw.get("one_launch")) and not kw.get("fast_fir") and kw["rrc"] == "wide" and kw["sps"] == 10
        bounded = f16 or (dom_name == "k_chain" and not kw.get("fast_fir") and kw["rrc"] == "narrow")
        ceiling, ceiling_rates = copy_ceiling(self.torch, p["x"].device, p["eng"].ctx)
        if f16:
            what = ("vector instruction issue: the 81-tap FIR runs as a split-f16 product on the matrix cores (36 v_mfma_f32_16x16x32_f16 per 1024 "
                    "outputs, error radius carried, undecided comparisons re-evaluated with the reference's rounded arithmetic: bit-exact output); "
                    "what remains is the slicer's and decoder's vector / scalar work")
        else:
            what = "fp32 VALU (%d-tap FIR, %s)" % (taps, "FMA" if kw.get("fast_fir") else
                                                   "FMA with a proven error radius, undecided comparisons re-evaluated with the reference's rounded arithmetic: bit-exact output"
                                                   if bounded else "unfused mul+add for bit-exactness")
        co = {"what": what, "fir_useful_tflops": fir_flops / (dom_ms * 1e-3) / 1e12, "peak_tflops_f32_vector": 157.3}
        for k in ("valu_issue_frac", "mfma_busy_frac", "sq_insts_valu_per_launch", "sq_insts_salu_per_launch", "sq_insts_mfma_per_launch"):
            if k in pc:
                co[k] = pc[k]
        if "sq_insts_valu_per_launch" in pc and kw.get("demod", "gfsk") != "none":
            # What binds (the reported roofline is HBM because the metric asks for it; HBM is idle most of the launch): the instructions a
            # wavefront issues per RUN of the slicer (one variance block: 100 symbols = 100 x sps samples of one channel), by kind,
            # from the committed counter passes scaled to this launch, and the SIMD cycles a run costs.
            runs = B * T / (100.0 * kw["sps"])
            issue = {"unit": "instructions per run (100 symbols of one channel: %d samples)" % (100 * kw["sps"]), "runs_per_launch": runs}
            for name_, key in (("vector", "sq_insts_valu_per_launch"), ("scalar", "sq_insts_salu_per_launch"), ("lds", "sq_insts_lds_per_launch"),
                               ("mfma", "sq_insts_mfma_per_launch"), ("branch", "sq_insts_branch_per_launch"),
                               ("vmem_read", "sq_insts_vmem_rd_per_launch"), ("vmem_write", "sq_insts_vmem_wr_per_launch"),
                               ("lds_bank_conflict_cycles", "sq_lds_bank_conflict_per_launch"), ("lds_active_cycles", "sq_lds_idx_active_per_launch")):
                if key in pc:
                    issue[name_] = pc[key] / runs
            # 1 024 SIMDs share the launch: SIMD-cycles per run = launch duration x shader clock x 1 024 / runs.  The clock is the
            # counter pass's own: GRBM_GUI_ACTIVE cycles of the launch (summed over the 8 XCDs) over the duration that pass's bench
            # line reported for it (tools/profile_gpu.sh writes it into the `## config:` line); passes older than that give cycles
            # only, and the clock is then those cycles over THIS run's duration (same kernel, another lease: marked as such).
            if "gpu_cycles_per_launch_profiled" in pc:
                prof_ms = pc.get("profiled_avg_launch_ms")
                clock_ghz = pc["gpu_cycles_per_launch_profiled"] / ((prof_ms * pc.get("profile_scale", 1.0) if prof_ms else dom_ms) * 1e6)
                issue["shader_clock_source"] = "GRBM_GUI_ACTIVE / 8 over " + ("the counter pass's own launch duration" if prof_ms else "this run's launch duration")
            else:
                clock_ghz = None
            issue["shader_clock_ghz"] = clock_ghz
            if clock_ghz:
                issue["simd_cycles_per_run"] = dom_ms * 1e-3 * clock_ghz * 1e9 * 1024.0 / runs
            if "vector" in issue and "simd_cycles_per_run" in issue:
                issue["simd_cycles_per_vector_instruction"] = issue["simd_cycles_per_run"] / issue["vector"]
            issue["microbench_floor"] = {"source": "tools/microbench/clock_probe.hip, valu_rate.hip (MI355X, four wavefronts per SIMD)",
                                         "cycles_per_plain_f32_or_int_vector_instruction": 2.4, "cycles_per_packed_conversion_minmax_dpp_instruction": 4.2,
                                         "cycles_per_v_fma_mix": 8.2, "cycles_per_mfma_16x16x32_f16": 16.0, "cycles_per_dependent_scalar_pair": 12.75,
                                         "one_wavefront_cycles_per_instruction_any_kind": 5.3}
            issue["source"] = pc.get("pmc_source")
            co["issue"] = issue
        return {"bound": "hbm", "kernel": dom_name + ("<%s>" % kw["proto"] if dom_name == "k_chain" else "") + (" PART 0" if group else ""),
                "launch_group": group,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "peak_achievable": ceiling, "frac_of_achievable": achieved / ceiling,
                "peak_achievable_source": "streaming rates over 2 GiB timed on this lease, the largest of: the library's own "
                                          "16-byte-per-lane non-temporal copy kernel (dh_debug_copy), torch.Tensor.copy_ and the library's read-only stream (dh_debug_copy, null destination: read bytes only)",
                "peak_achievable_rates": ceiling_rates,
                "traffic": pc.get("traffic"), "traffic_source": pc.get("traffic_source"),
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_step": pc.get("algorithmic_bytes_per_step"), "avg_launch_ms": dom_ms,
                "co_limit": co}, \
               {"rrc": mean(rrc_ms), "slicer": mean(slicer_ms), "decoder": mean(dec_ms)}

    def verify_timed(self, nv, pushes):
        """What the TIMED engines produced in their last push -- the engines of Job.timed themselves, with whatever launch
        path they took (tail split, overlapped pushes, two streams) -- for nv channels spread over every part, against the
        oracle: the same rows as one stream of `pushes` identical pushes through the oracle, and of pushes - 1; what the
        longer run has beyond the shorter one is the last push's output (dibits, decoder bytes, events, filtered floats).
        Outside any timed region."""
        import numpy as np
        from oracle import oracle as O
        if any("iq" in p for p in self.parts):
            return None, None                       # (the front-end's state makes every push's floats different: Job.verify replays those)
        ok, frame_bytes, worst, checked = True, 0, 0.0, 0
        dibits_ok = True
        for p in self.parts:
            kw, T, eng = p["kw"], p["T"], p["eng"]
            n = min(nv, p["B"])
            pick = [int(round(v)) for v in np.linspace(0, p["B"] - 1, n)]
            xs = p["x"][self.torch.tensor(pick, device=p["x"].device)].cpu().numpy()
            okw = oracle_kw(p["proto"])
            if kw["proto"] == "none":
                okw = dict(okw, proto=0)
            threads = min(n, effective_cores()[1])
            keep = bool(kw.get("keep_filtered"))
            full = O.chain(np.tile(xs, (1, pushes)), threads=threads, keep_filtered=keep, **okw)
            pre = O.chain(np.tile(xs, (1, pushes - 1)), threads=threads, **okw) if pushes > 1 else None
            lo = (lambda key, b: int(pre[key][b])) if pre is not None else (lambda key, b: 0)
            gs, gsc = eng.read_rows("symbols", pick)
            if kw["proto"] != "none":
                gf, gfc = eng.read_rows("frames", pick)
                ge, gec = eng.read_rows("events", pick)
            for j in range(n):
                want = full["syms"][j, lo("sym_count", j):full["sym_count"][j]]
                same = gsc[j] == len(want) and gs[j, :gsc[j]].tobytes() == want.tobytes()
                if kw.get("fast_fir") and not kw.get("one_launch"):
                    same = True                      # dibits are not guaranteed with the FMA FIR of the two-kernel pair; the floats are checked below
                dibits_ok = dibits_ok and bool(same)
                ok &= bool(same)
                if kw["proto"] != "none":
                    wf = full["out"][j, lo("out_count", j):full["out_count"][j]]
                    we = full["events"][j, lo("event_count", j):full["event_count"][j]]
                    ok &= bool(gfc[j] == len(wf) and gf[j, :gfc[j]].tobytes() == wf.tobytes())
                    if not kw.get("fast_fir"):
                        # (event positions count from the start of the stream on both sides)
                        ok &= bool(gec[j] == len(we) and ge[j, :gec[j]].tobytes() == we.tobytes())
                    frame_bytes += int(gfc[j])
            if keep:
                y = eng.read_rows("filtered", pick)[0][:, :T]
                r = full["filtered"][:, (pushes - 1) * T:]
                if kw.get("fast_fir") or kw.get("one_launch"):               # 1e-6 relative to max(|ref|, rms(ref)); DH_FLAG_ONE_LAUNCH promises 2.5e-6
                    rms = np.sqrt(np.mean(r.astype(np.float64) ** 2)) + 1e-30
                    err = float(np.max(np.abs(y.astype(np.float64) - r) / np.maximum(np.abs(r), rms)))
                    worst = max(worst, err)
                    ok &= err <= (2.5e-6 if kw.get("one_launch") and not kw.get("fast_fir") else 1e-6)
                else:
                    ok &= bool((y.view(np.uint32) == r.view(np.uint32)).all())
            checked += n
        out = {"what": "the timed engines' own last push (push %d of %d identical pushes since reset) against the oracle run over the same stream" % (pushes, pushes),
               "channels": checked, "sampling": "evenly spread over the batch", "pushes": pushes,
               "bit_exact_vs_oracle": bool(ok) and not self.kw.get("fast_fir") and not self.kw.get("one_launch"), "frame_bytes": frame_bytes}
        if self.kw.get("fast_fir"):
            out.update({"within_1e-6_vs_oracle": bool(ok), "max_rel_err": worst})
        if self.kw.get("one_launch"):
            out.update({"dibits_bit_exact_vs_oracle": bool(dibits_ok), "max_rel_err": worst, "within_1e-6_vs_oracle": bool(ok and worst <= 1e-6)})
            if not self.kw.get("fast_fir"):
                out["floats_within_2.5e-6_vs_oracle"] = bool(ok)
        return bool(ok), out

    def verify(self, ctx, nv, reps=2):
        """Replay the first nv channels of every part from reset on a small engine and on the oracle (outside any timed
        region): dibits, decoder bytes (and the filtered floats of the materialised-RRC configs) must agree."""
        import numpy as np
        from digiham_amd import api
        from oracle import oracle as O
        ok, frame_bytes, worst = True, 0, 0.0
        for p in self.parts:
            kw, T = p["kw"], p["T"]
            n = min(nv, p["B"])
            small = api.Engine(n, T, ctx=ctx, **kw)
            pick = self.torch.linspace(0, p["B"] - 1, n, device=p["x"].device).round().long()      # spread over the whole batch
            xs = p["x"][pick].contiguous()
            got_s, got_f, got_y = [[] for _ in range(n)], [[] for _ in range(n)], []
            for _ in range(reps):
                small.push(xs)
                s, sc = small.symbols()
                for b in range(n):
                    got_s[b].append(s[b, :sc[b]].copy())
                if kw["proto"] != "none":
                    f, fc = small.frames()
                    for b in range(n):
                        got_f[b].append(f[b, :fc[b]].copy())
                if kw.get("keep_filtered"):
                    got_y.append(small.filtered()[:, :T].copy())
            small.close()
            xh = np.tile(xs.cpu().numpy(), (1, reps))
            okw = oracle_kw(p["proto"])
            if kw["proto"] == "none":
                okw = dict(okw, proto=0)
            ref = O.chain(xh, threads=min(n, effective_cores()[1]), keep_filtered=bool(kw.get("keep_filtered")), **okw)
            for b in range(n):
                gs = np.concatenate(got_s[b])
                same = len(gs) == ref["sym_count"][b] and bool((gs == ref["syms"][b, :len(gs)]).all())
                if kw.get("fast_fir") and not kw.get("one_launch"):
                    same = True                      # dibits are not guaranteed with the FMA FIR of the two-kernel pair; the floats are checked below
                ok &= same
                if kw["proto"] != "none":
                    gf = np.concatenate(got_f[b])
                    ok &= len(gf) == ref["out_count"][b] and bool((gf == ref["out"][b, :len(gf)]).all())
                    frame_bytes += len(gf)
            if kw.get("keep_filtered"):
                y, r = np.concatenate(got_y, axis=1), ref["filtered"]
                if kw.get("fast_fir") or kw.get("one_launch"):               # 1e-6 relative to max(|ref|, rms(ref)); DH_FLAG_ONE_LAUNCH promises 2.5e-6
                    rms = np.sqrt(np.mean(r.astype(np.float64) ** 2)) + 1e-30
                    err = float(np.max(np.abs(y.astype(np.float64) - r) / np.maximum(np.abs(r), rms)))
                    worst = max(worst, err)
                    ok &= err <= (2.5e-6 if kw.get("one_launch") and not kw.get("fast_fir") else 1e-6)
                else:
                    ok &= bool((y.view(np.uint32) == r.view(np.uint32)).all())
        out = {"channels": nv, "sampling": "evenly spread over the batch", "pushes": reps,
               "bit_exact_vs_oracle": bool(ok) and not self.kw.get("fast_fir") and not self.kw.get("one_launch"), "frame_bytes": frame_bytes}
        if self.kw.get("fast_fir"):
            out.update({"within_1e-6_vs_oracle": bool(ok), "max_rel_err": worst})
        if self.kw.get("one_launch"):
            out.update({"dibits_bit_exact_vs_oracle": bool(ok), "max_rel_err": worst, "within_1e-6_vs_oracle": bool(ok and worst <= 1e-6)})
            if not self.kw.get("fast_fir"):
                out["floats_within_2.5e-6_vs_oracle"] = bool(ok)
        return ok, out


def other_configs(torch, ctx, device, steps, warmup, verify):
    """The remaining single-GPU BASELINE configs on the same lease (each its own engines, inputs resident, same timing
    method as the headline)."""
    out = []
    # ("mixed", (4096, 4096)) is one GPU's share of BASELINE configs[4] (65 536 channels over 8 GPUs), ("dmr_full", 8192) its
    # share of the north-star target (65 536 DMR channels over 8 GPUs)
    for workload, channels, overlap, streams in (("rrc_gfsk", 4096, False, 1), ("rrc_gfsk_fast", 4096, False, 1), ("rrc_gfsk_one", 4096, False, 1), ("rrc_gfsk_fast_one", 4096, False, 1), ("ysf_full", 16384, False, 1),
                                                 ("mixed", (8192, 8192), False, 1), ("mixed", (8192, 8192), False, 2),
                                                 ("mixed", (4096, 4096), False, 1), ("mixed", (4096, 4096), False, 2), ("dmr_full", 8192, False, 1),
                                                 ("dmr_full", 16384, True, 1), ("ysf_full", 16384, True, 1),
                                                 ("dmr_iq_full", 16384, False, 1)):       # (`--workload dmr_iq_full --streams 2`, the front-end on its own stream, gains 3 %: both kernels fill the chip)
        t_start = time.perf_counter()
        job = Job(torch, ctx, device, workload, channels, rank=0, overlap=overlap, streams=streams)
        dt = job.timed(steps, warmup)
        roof, stage = job.roofline(step_ms=dt / steps * 1e3)
        step_alg = job.step_algorithmic_bytes()
        entry = {"workload": workload + (" --overlap" if overlap else "") + (" --streams 2" if streams == 2 else ""),
                 "channels": channels if isinstance(channels, int) else list(channels),
                 # the whole step's algorithmic bytes over the step period where the step is more than the dominant launch
                 "algorithmic_bytes_per_whole_step": step_alg, "frac_step": step_alg / (dt / steps) / 1e9 / HBM_PEAK_GBS,
                 "config": "%s channels x %d samples: %s%s" % (channels, job.parts[0]["T"], job.desc,
                                                              "; pushes overlapped on the engine's own streams (DH_FLAG_OVERLAP_PUSHES)" if overlap else ""),
                 "launch_group": roof.get("launch_group"),
                 "steps": steps, "ms_per_step": dt / steps * 1e3, "value": job.samples_per_step * steps / dt / SAMPLE_RATE, "unit": "channels",
                 "kernel": roof["kernel"], "avg_launch_ms": roof["avg_launch_ms"], "frac": roof["frac"], "frac_of_achievable": roof["frac_of_achievable"],
                 "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"], "algorithmic_bytes_per_step": roof.get("algorithmic_bytes_per_step"),
                 "traffic": roof["traffic"], "traffic_source": roof["traffic_source"],
                 "stage_ms": stage}
        if verify:
            ok, entry["verified"] = job.verify_timed(verify, steps + warmup)
            if ok is None:
                ok, entry["verified"] = job.verify(ctx, verify)          # front-end workloads: a replay on a fresh small engine
            assert ok, "GPU output of the timed engines differs from the oracle (%s)" % entry["workload"]
        job.close()
        del job
        torch.cuda.empty_cache()
        entry["wall_s"] = time.perf_counter() - t_start
        out.append(entry)
    return out


MAX_LINE_BYTES = 4096          # the driver keeps an 8 KB tail of stdout and parses its last line: round 5's 20.7 KB line was lost


def _sig(v, digits=6):
    """Numbers of the final line at `digits` significant digits (floats that are whole numbers stay ints)."""
    if isinstance(v, bool) or v is None or isinstance(v, (str, int)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        if v == int(v) and abs(v) < 1e15:
            return int(v)
        return float("%.*g" % (digits, v))
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def compact_line(line, detail_path=None):
    """The LAST stdout line: the contract's keys, `roofline`, `cpu_baseline`, `verified` and one short record per other
    workload -- numbers only, no prose.  Everything else (sources, notes, reference_fec, rates, the micro-benchmark floor)
    stays in the full record: bench_detail.json beside bench.py and the `BENCH_DETAIL ` stdout line printed before this one."""
    pick = lambda d, keys: {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    out = pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = line.get("config", {})
    out["config"] = pick(cfg, ("channels_per_gpu", "samples_per_channel_per_step", "total_channels", "sharding"))
    out["config"]["workload"] = str(cfg.get("workload", ""))[:160]
    out["msamples_per_s"] = line.get("msamples_per_s")
    if line.get("frac_step") is not None:
        out["frac_step"] = line["frac_step"]
    roof = line.get("roofline") or {}
    r = pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "peak_achievable", "frac_of_achievable", "traffic",
                    "algorithmic_bytes_per_launch", "avg_launch_ms"))
    co = roof.get("co_limit") or {}
    issue = co.get("issue") or {}
    if issue:
        # `bound` stays "hbm" (the roofline the metric is quoted against); what binds is said here
        r["co_limit"] = {"binds": "instruction issue", "valu_issue_frac": co.get("valu_issue_frac"), "mfma_busy_frac": co.get("mfma_busy_frac"),
                         "issue": pick(issue, ("vector", "scalar", "lds", "mfma", "branch", "vmem_read", "vmem_write", "lds_bank_conflict_cycles",
                                               "lds_active_cycles", "simd_cycles_per_run", "shader_clock_ghz"))}
    out["roofline"] = r
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = pick(cb, ("value", "unit", "msamples_per_s", "cores", "kind", "gpu_matches_baseline_outputs", "error"))
        if "sample" in cb:
            out["cpu_baseline"]["sample"] = "%s channels x %s samples, oracle/ port" % (cb.get("channels_hashed"), cfg.get("samples_per_channel_per_step"))
        rf = cb.get("reference_fec")
        if isinstance(rf, dict) and "bptc_196_96" in rf:
            out["cpu_baseline"]["reference_fec"] = {k: {"ref_per_s": rf[k]["reference_blocks_per_s"], "gpu_per_s": rf[k]["gpu_blocks_per_s"], "identical": rf[k]["identical"]}
                                                    for k in ("bptc_196_96", "trellis_180") if k in rf}
    v = line.get("verified")
    if isinstance(v, dict):
        out["verified"] = pick(v, ("bit_exact_vs_oracle", "channels", "pushes", "within_1e-6_vs_oracle", "dibits_bit_exact_vs_oracle", "max_rel_err"))
    oc = line.get("other_configs")
    if isinstance(oc, list):
        rows = []
        for e in oc:
            ver = e.get("verified") or {}
            ok = ver.get("bit_exact_vs_oracle")
            if not ok and ("within_1e-6_vs_oracle" in ver or "floats_within_2.5e-6_vs_oracle" in ver):
                # the float-tolerance workloads: what each promises (2.5e-6 for the split-f16 one-launch mode, 1e-6 otherwise; dibits where guaranteed)
                ok = bool(ver.get("floats_within_2.5e-6_vs_oracle", ver.get("within_1e-6_vs_oracle"))) and ver.get("dibits_bit_exact_vs_oracle", True)
            row = {"workload": e.get("workload"), "channels": e.get("channels"), "ms_per_step": e.get("ms_per_step"), "frac": e.get("frac"),
                   "frac_step": e.get("frac_step"), "ok": ok}
            alg = e.get("algorithmic_bytes_per_step") or e.get("algorithmic_bytes_per_launch")
            if e.get("traffic") and alg:
                row["traffic_ratio"] = e["traffic"] / alg
            if "max_rel_err" in ver:
                row["max_rel_err"] = ver["max_rel_err"]
            rows.append({k: v for k, v in row.items() if v is not None})
        out["other_configs"] = rows
    elif oc is not None:
        out["other_configs"] = oc
    if detail_path:
        out["detail"] = detail_path
    out = _sig(out)
    text = json.dumps(out, separators=(",", ":"))
    # belt and braces: a line over the bound loses the least important parts first, never the headline
    for drop in (("other_configs",), ("roofline", "co_limit"), ("cpu_baseline", "reference_fec"), ("config", "workload")):
        if len(text) <= MAX_LINE_BYTES:
            break
        d = out
        for k in drop[:-1]:
            d = d.get(k, {})
        if drop[-1] in d:
            d[drop[-1]] = "see " + (detail_path or "the BENCH_DETAIL line")
        text = json.dumps(out, separators=(",", ":"))
    assert len(text) <= MAX_LINE_BYTES, "bench.py: final line %d bytes" % len(text)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dmr_full", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--channels", type=int, default=16384, help="channels per GPU (weak scaling)")
    ap.add_argument("--total-channels", type=int, default=65536, help="channels of the whole job (strong scaling)")
    ap.add_argument("--units", type=int, default=0, help="bursts (DMR, 30 ms) or frames (YSF, 100 ms) per step; 0 = ~4 s")
    ap.add_argument("--streams", type=int, default=1, help="mixed: 2 = the two engines on their own HIP streams; dmr_iq_full: 2 = the front-end of the next step on its own stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--split-stages", action="store_true", help="slicer and decoder as two kernels (per-stage timing)")
    ap.add_argument("--overlap", action="store_true", help="DH_FLAG_OVERLAP_PUSHES: pushes as two launches on the engine's own streams, joined at the end")
    ap.add_argument("--verify", type=int, default=64, help="channels checked bit-exact against the oracle after the run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)            # does not return

    import numpy as np
    import torch
    from digiham_amd import api, shard

    rank, world, local = shard.init_process_group()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s): launch with --nproc-per-node == --gpus" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs; there is no CPU path"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    ctx = api.Context(device=local)

    mixed = args.workload == "mixed"
    if args.scaling == "strong":                # a fixed job, split by channel index
        if mixed:
            half = args.total_channels // 2
            (a0, a1), (b0, b1) = shard.channel_range(half, rank, world), shard.channel_range(args.total_channels - half, rank, world)
            channels = (a1 - a0, b1 - b0)
        else:
            lo, hi = shard.channel_range(args.total_channels, rank, world)
            channels = hi - lo
    else:
        channels = (args.channels // 2, args.channels - args.channels // 2) if mixed else args.channels
    job = Job(torch, ctx, device, args.workload, channels, rank, split_stages=args.split_stages, streams=args.streams, units=args.units, overlap=args.overlap)

    dt = job.timed(args.steps, args.warmup, barrier=shard.barrier)
    samples = job.samples_per_step * args.steps
    dt_max, samples_all = shard.reduce_report(dt, samples, device)
    roof, stage = job.roofline(args.split_stages, step_ms=dt / args.steps * 1e3)

    verified = None
    if args.verify:
        ok, verified = job.verify_timed(args.verify, args.steps + args.warmup)
        if ok is None and not (job.kw["proto"] == "none" and not job.kw.get("keep_filtered")):
            ok, verified = job.verify(ctx, args.verify)          # front-end workloads: a replay on a fresh small engine
        assert ok is not False, "GPU output differs from the oracle"

    if rank == 0:
        rate = samples_all / dt_max
        proto = job.proto
        per_gpu = [p["B"] for p in job.parts]
        T = job.parts[0]["T"]
        line = {
            "metric": "concurrent 48 kS/s DMR+YSF channels sustained end-to-end" if mixed else
                      "concurrent 48 kS/s DMR channels sustained end-to-end (rrc_filter->gfsk_demodulator->dmr_decoder)"
                      if proto == "dmr" else "concurrent 48 kS/s %s channels sustained end-to-end" % proto.upper(),
            "value": rate / SAMPLE_RATE, "unit": "channels",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %s channels%s x %.2f s (%d samples) of 48 kS/s FM-discriminator audio, %s"
                                   % ("+".join(str(b) for b in per_gpu), "DMR+YSF" if mixed else proto.upper(),
                                      "/GPU" if args.scaling == "weak" else " on rank 0 of a fixed %d-channel job" % args.total_channels,
                                      T / SAMPLE_RATE, T, job.desc),
                       "channels_per_gpu": sum(per_gpu), "samples_per_channel_per_step": T,
                       "total_channels": int(round(samples_all / args.steps / T)) if not mixed else
                                         (args.total_channels if args.scaling == "strong" else sum(per_gpu) * world),
                       "sharding": "channels, no collective", "streams": args.streams, "overlap_pushes": bool(args.overlap),
                       # chain launches of >= 8192 channels x >= 65536 samples: two workgroups per channel in ONE launch, the second takes
                       # the rows from this percentage on (engine.hip: go_chain / k_chain); "0" = one workgroup per channel
                       "tail_split_pct": os.environ.get("DH_TAIL_SPLIT", "80")},
            "msamples_per_s": rate / 1e6,
            "frac_step": job.step_algorithmic_bytes() * args.steps / dt / 1e9 / HBM_PEAK_GBS,      # this rank's whole step over its own period
            "roofline": roof, "stage_ms": stage, "verified": verified,
        }
        if world == 1 and not args.no_cpu_baseline:
            x0 = job.parts[0]["x"]

            def host_rows(k):
                return np.ascontiguousarray(x0[:k].cpu().numpy())
            try:
                line["cpu_baseline"] = cb = cpu_baseline(host_rows, proto)
            except Exception as e:          # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = cb = {"error": repr(e)}
            if "outputs_sha256" in cb and job.kw["proto"] != "none":
                # the same channels through a fresh engine of the product: every byte the oracle run of the baseline produced
                k = cb["channels_hashed"]
                eng = api.Engine(k, T, ctx=ctx, **job.kw)
                eng.push(x0[:k].contiguous())
                gs, gsc = eng.symbols()
                gf, gfc = eng.frames()
                eng.close()
                cb["gpu_outputs_sha256"] = outputs_sha(gs, gsc, gf, gfc)
                cb["gpu_matches_baseline_outputs"] = cb["gpu_outputs_sha256"] == cb["outputs_sha256"]
                assert cb["gpu_matches_baseline_outputs"], "GPU output differs from the oracle run of the CPU baseline"
            try:
                cb["reference_fec"] = reference_fec_baseline(torch, ctx)
            except Exception as e:
                cb["reference_fec"] = {"error": repr(e)}
    job.close()
    del job
    if rank == 0:
        default_headline = args.workload == "dmr_full" and args.scaling == "weak" and args.channels == 16384
        if world == 1 and default_headline and not args.no_other_configs:
            torch.cuda.empty_cache()
            try:
                line["other_configs"] = other_configs(torch, ctx, device, min(args.steps, 10), min(args.warmup, 2), args.verify)
            except Exception as e:
                line["other_configs"] = {"error": repr(e)}
        assert line["n_gpus"] == args.gpus
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        shard.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        try:                                                        # RCCL's own lines ("Hostname", "Librccl path") sit in the C library's stdout buffer
            import ctypes                                           # until exit: out with them now, so that the JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        detail = None
        for cand in (os.environ.get("DH_BENCH_DETAIL"), os.path.join(ROOT, "bench_detail.json")):
            if not cand:
                continue
            try:
                with open(cand, "w") as f:
                    json.dump(line, f, indent=1)
                detail = os.path.relpath(cand, ROOT) if cand.startswith(ROOT) else cand
                break
            except OSError:
                continue
        print("BENCH_DETAIL " + json.dumps(line), flush=True)          # the full record (sources, notes, per-workload detail): NOT the line the driver parses
        print(compact_line(line, detail), flush=True)                  # the last line of stdout, after RCCL's own chatter: <= 4 KB


if __name__ == "__main__":
    main()
