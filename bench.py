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

One JSON line on rank 0: value = whole-job real-time 48 kS/s channels = samples/s / 48 000, plus
`roofline` for the dominant kernel (k_chain, timed with HIP events on its own stream inside
the timed region) and `cpu_baseline` (the oracle's scalar restatement of the reference pipe on
this box's host cores, bounded sample, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable with a copy kernel
SAMPLE_RATE = 48000

WORKLOADS = {
    # name: (proto, engine kwargs, algorithmic bytes per input sample of the dominant kernel, description)
    "dmr_full": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="dmr"),
                 "full chain rrc(wide)->gfsk(10)->dmr_decoder incl. BPTC(196,96) (BASELINE configs[2])"),
    "ysf_full": ("ysf", dict(rrc="wide", demod="gfsk", sps=10, proto="ysf"),
                 "full chain rrc(wide)->gfsk(10)->ysf_decoder incl. Golay/Viterbi (BASELINE configs[3])"),
    "rrc_gfsk": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="none", keep_filtered=True),
                 "rrc(wide) materialised + gfsk(10), float path (BASELINE configs[1])"),
    "rrc_gfsk_fast": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="none", keep_filtered=True, fast_fir=True),
                      "rrc(wide) materialised with the FMA FIR (1e-6 float tolerance of configs[1]) + gfsk(10)"),
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
    # BASELINE configs[4] per GPU: half the channels DMR, half YSF, one engine (and one launch per push) each
    "mixed": ("dmr", dict(rrc="wide", demod="gfsk", sps=10, proto="dmr"),
              "half DMR + half YSF channels, full chains (BASELINE configs[4] per-GPU share)"),
}


def profiled_traffic(workload, channels, T):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE,
    KiB, per-dispatch average; MI355X_MICROARCH.md) -- bench.py cannot collect counters itself.  Only for the
    configuration those passes were run on (tools/profile_gpu.sh: the default workload)."""
    path = os.path.join(ROOT, "profiles", "r01_k_chain_pmc.txt")
    if workload != "dmr_full" or channels != 16384 or T != 190080 or not os.path.exists(path):
        return None, None
    fetch = write = None
    for line in open(path):
        if "k_chain" in line and " FETCH_SIZE " in line:
            fetch = float(line.split("avg=")[1].split()[0])
        if "k_chain" in line and " WRITE_SIZE " in line:
            write = float(line.split("avg=")[1].split()[0])
    if fetch is None or write is None:
        return None, None
    return fetch * 2.0 * 1024.0 + write * 1024.0, "profiles/r01_k_chain_pmc.txt (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of this workload)"


def oracle_kw(proto):
    if proto == "dstar":
        return dict(proto=5, rrc=0, levels=2, sps=10)
    if proto == "pocsag":
        return dict(proto=4, rrc=0, levels=2, sps=40, invert=True)
    return dict(proto={"dmr": 1, "ysf": 2, "nxdn": 3}[proto], **(dict(rrc=2, sps=20) if proto == "nxdn" else {}))


def cpu_baseline(x_host_fn, proto, budget_s=12.0):
    """Time the oracle (scalar restatement of the reference pipe) on this box's host cores."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    probe = x_host_fn(1)
    n = probe.shape[1]
    t0 = time.perf_counter()
    O.chain(probe[:, : min(n, 96000)], threads=1, **oracle_kw(proto))
    per_sample = (time.perf_counter() - t0) / min(n, 96000)
    # size the sample for ~budget_s of wall time with every core busy
    chans = max(cores, int(budget_s / (per_sample * n) * cores))
    chans = min(chans, 16384)
    x = x_host_fn(chans)
    t0 = time.perf_counter()
    O.chain(x, threads=cores, **oracle_kw(proto))
    dt = time.perf_counter() - t0
    rate = x.size / dt
    return {"value": rate / SAMPLE_RATE, "unit": "channels", "msamples_per_s": rate / 1e6,
            "msamples_per_s_per_core": rate / 1e6 / cores, "single_thread_msamples_per_s": 1e-6 / per_sample,
            "cores": cores, "kind": "port",
            "sample": "%d channels x %d samples of the same synthetic workload through oracle/ (scalar C restatement of "
                      "[rrc_filter|]g/fsk_demodulator|%s_decoder, bit-exact with the GPU path), %d pthreads, %.1f s wall"
                      % (chans, x.shape[1], proto, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dmr_full", choices=sorted(WORKLOADS))
    ap.add_argument("--channels", type=int, default=16384, help="channels per GPU (weak scaling)")
    ap.add_argument("--units", type=int, default=0, help="bursts (DMR, 30 ms) or frames (YSF, 100 ms) per step; 0 = ~4 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split-stages", action="store_true", help="slicer and decoder as two kernels (per-stage timing)")
    ap.add_argument("--verify", type=int, default=8, help="channels checked bit-exact against the oracle after the run")
    args = ap.parse_args()

    import numpy as np
    import torch
    from digiham_amd import api, shard, synth_torch

    rank, world, local = shard.init_process_group()
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs; there is no CPU path"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    proto, kw, desc = WORKLOADS[args.workload]
    B = args.channels
    units = args.units or {"dmr": 132, "ysf": 40, "nxdn": 50, "dstar": 198, "pocsag": 148}[proto]
    x, info = synth_torch.make_batch(torch, device, proto, B, units, seed=1000 + 7919 * rank, sps=kw["sps"])
    T = info["samples_per_channel"]
    ctx = api.Context(device=local)
    if args.split_stages:
        kw = dict(kw, split_stages=True)
    mixed = args.workload == "mixed"
    if mixed:
        B = B // 2                          # this many DMR channels + as many YSF channels
        x = x[:B].contiguous()
        x2, info2 = synth_torch.make_batch(torch, device, "ysf", B, 40, seed=2000 + 7919 * rank)
        T2 = info2["samples_per_channel"]
        eng2 = api.Engine(B, T2, ctx=ctx, **dict(kw, proto="ysf"))
    eng = api.Engine(B, T, ctx=ctx, **kw)
    n_timed = args.steps
    eng.timing_enable(max(n_timed, 1))

    def step():
        eng.push(x)
        if mixed:
            eng2.push(x2)

    for _ in range(args.warmup):
        step()
    eng.sync()
    eng.timing_read()                       # drop warm-up timings
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    shard.barrier()
    dt = time.perf_counter() - t0
    eng.sync()                              # raises on any output-buffer overflow
    if mixed:
        eng2.sync()
    rrc_ms, slicer_ms, dec_ms = eng.timing_read()
    frame_bytes_step = 0
    if kw["proto"] != "none":
        frame_bytes_step = int(eng.frames()[1].sum())          # decoder output of the last step, all channels

    samples = float(B) * T * args.steps + (float(B) * T2 * args.steps if mixed else 0.0)
    dt_max, samples_all = shard.reduce_report(dt, samples, device)

    # ---- parity spot check on the last step (outside the timed region)
    verified = None
    if args.verify and kw["proto"] != "none" and not kw.get("fast_fir"):
        from oracle import oracle as O
        nv = min(args.verify, B)
        # replay the whole stream of the first nv channels from reset on a small engine, and on the oracle
        small = api.Engine(nv, T, ctx=ctx, **kw)
        xs = x[:nv].contiguous()
        got_s, got_f = [[] for _ in range(nv)], [[] for _ in range(nv)]
        reps = 2
        for _ in range(reps):
            small.push(xs)
            s, sc = small.symbols()
            f, fc = small.frames()
            for b in range(nv):
                got_s[b].append(s[b, :sc[b]].copy()); got_f[b].append(f[b, :fc[b]].copy())
        xh = np.tile(xs.cpu().numpy(), (1, reps))
        ref = O.chain(xh, threads=min(nv, os.cpu_count() or 1), **oracle_kw(proto))
        ok = True
        for b in range(nv):
            gs, gf = np.concatenate(got_s[b]), np.concatenate(got_f[b])
            ok &= len(gs) == ref["sym_count"][b] and bool((gs == ref["syms"][b, :len(gs)]).all())
            ok &= len(gf) == ref["out_count"][b] and bool((gf == ref["out"][b, :len(gf)]).all())
        verified = {"channels": nv, "pushes": reps, "bit_exact_vs_oracle": bool(ok),
                    "frame_bytes": int(sum(len(np.concatenate(g)) for g in got_f))}
        small.close()
        assert ok, "GPU output differs from the oracle"

    if rank == 0:
        rate = samples_all / dt_max
        n_gpus = world
        # dominant kernel: fused RRC + slicer (k_rrc_demod); algorithmic bytes per launch =
        # input f32 (4 B/sample) + dibits out (1 B per 10 samples) -- SURVEY.md section 8(d)
        alg_bytes = B * T * 4.0 + B * (T / float(kw["sps"]))   # of ONE launch of the dominant kernel (mixed: the DMR engine's)
        if kw.get("keep_filtered"):
            # unfused config 2: the RRC kernel is dominant; 4 B in + 4 B out per sample
            dom_ms = float(np.mean(rrc_ms)) if len(rrc_ms) else float("nan")
            alg_bytes = B * T * 8.0
            dom_name = "k_rrc_tile"
        else:
            dom_ms = float(np.mean(slicer_ms)) if len(slicer_ms) else float("nan")
            # one launch for slicer + decoder exists for the sps-10 DMR / YSF chains (engine.hip: launch_chain)
            chained = ((kw["proto"] in ("dmr", "ysf") or (kw["proto"] == "dstar" and kw["rrc"] == "none")) and kw["sps"] == 10
                       or (kw["proto"] == "nxdn" and kw["rrc"] == "narrow")) and not args.split_stages
            dom_name = "k_chain" if chained else "k_rrc_demod"
            if chained:
                alg_bytes += frame_bytes_step          # + decoder output (<= 27 B per 1440 samples for DMR)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_src = profiled_traffic(args.workload, B, T) if dom_name == "k_chain" else (None, None)
        taps = {"wide": 81, "narrow": 161, "none": 0}[kw["rrc"]]
        fir_flops = B * T * 2.0 * taps       # one mul + one add per tap and sample, unfused
        line = {
            "metric": "concurrent 48 kS/s DMR+YSF channels sustained end-to-end" if mixed else
                      "concurrent 48 kS/s DMR channels sustained end-to-end (rrc_filter->gfsk_demodulator->dmr_decoder)"
                      if proto == "dmr" else "concurrent 48 kS/s %s channels sustained end-to-end" % proto.upper(),
            "value": rate / SAMPLE_RATE, "unit": "channels",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d %s channels/GPU x %.2f s (%d samples) of 48 kS/s FM-discriminator audio, %s"
                                   % (2 * B if mixed else B, "DMR+YSF" if mixed else proto.upper(), T / SAMPLE_RATE, T, desc),
                       "channels_per_gpu": B, "samples_per_channel_per_step": T, "sharding": "channels, no collective"},
            "msamples_per_s": rate / 1e6,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms,
                         "co_limit": {"what": "fp32 VALU (%d-tap FIR, unfused mul+add for bit-exactness)" % taps,
                                      "achieved_tflops": fir_flops / (dom_ms * 1e-3) / 1e12, "peak_tflops": 157.3}},
            "stage_ms": {"rrc": float(np.mean(rrc_ms)) if len(rrc_ms) else None,
                         "slicer": float(np.mean(slicer_ms)) if len(slicer_ms) else None,
                         "decoder": float(np.mean(dec_ms)) if len(dec_ms) else None},
            "verified": verified,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            def host_rows(k):
                return np.ascontiguousarray(x[:k].cpu().numpy())
            try:
                line["cpu_baseline"] = cpu_baseline(host_rows, proto)
            except Exception as e:          # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    eng.close()
    if mixed:
        eng2.close()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        shard.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
