"""-m gpu only: BASELINE-size batches checked through size-independent properties.

* replication: channels that carry the same samples must produce byte-identical outputs wherever they sit in
  the batch (4 096 / 16 384 channels are U distinct signals repeated) -- every channel is covered;
* a sampled subset is compared with the oracle;
* reset: the same push after dh_engine_reset gives the same outputs; continuing without reset does not.
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _digest(rows, counts):
    return [hashlib.sha256(rows[b, :counts[b]].tobytes()).hexdigest() for b in range(rows.shape[0])]


@pytest.mark.parametrize("proto,B,units", [("dmr", 16384, 33), ("ysf", 4096, 12), ("ysf", 16384, 12), ("dmr", 4096, 132), ("nxdn", 4096, 16), ("dstar", 4096, 60)])
def test_replicated_channels_agree_and_match_oracle(gpu_ctx, oracle, proto, B, units):
    import torch
    from digiham_amd import api, synth_torch
    U = 32
    kw = dict(rrc="narrow", sps=20) if proto == "nxdn" else dict(rrc="none", demod="fsk") if proto == "dstar" else {}
    base, info = synth_torch.make_batch(torch, gpu_ctx.mem.device, proto, U, units, U=U, seed=4242, sps=kw.get("sps", 10))
    T = info["samples_per_channel"]
    x = base.repeat(B // U, 1).contiguous()                      # channel ch carries signal ch % U
    eng = api.Engine(B, T, proto=proto, ctx=gpu_ctx, **kw)
    outs = []
    for _ in range(2):                                           # two pushes: state carry at scale
        eng.push(x)
        s, sc = eng.symbols()
        f, fc = eng.frames()
        e, ec = eng.events()
        outs.append((s, sc, f, fc, e, ec))
    eng.close()
    for s, sc, f, fc, e, ec in outs:
        for rows, counts in ((s, sc), (f, fc), (e.view(np.uint8).reshape(B, -1), ec * 32)):
            d = _digest(rows, counts)
            assert all(d[ch] == d[ch % U] for ch in range(B))
    # the U distinct signals against the oracle (both pushes as one stream)
    xh = np.tile(base.cpu().numpy(), (1, 2))
    okw = dict(rrc=2, sps=20) if proto == "nxdn" else dict(rrc=0, levels=2) if proto == "dstar" else {}
    ref = oracle.chain(xh, proto={"dmr": 1, "ysf": 2, "nxdn": 3, "dstar": 5}[proto], threads=8, **okw)
    for b in range(U):
        gs = np.concatenate([o[0][b, :o[1][b]] for o in outs])
        gf = np.concatenate([o[2][b, :o[3][b]] for o in outs])
        ge = np.concatenate([o[4][b, :o[5][b]] for o in outs])
        assert len(gs) == ref["sym_count"][b] and (gs == ref["syms"][b, :len(gs)]).all()
        assert len(gf) == ref["out_count"][b] and (gf == ref["out"][b, :len(gf)]).all()
        assert ge.tobytes() == ref["events"][b, :ref["event_count"][b]].tobytes()
    assert sum(int(o[3].sum()) for o in outs) > 0


def test_reset_restores_initial_state(gpu_ctx):
    import torch
    from digiham_amd import api, synth_torch
    x, info = synth_torch.make_batch(torch, gpu_ctx.mem.device, "dmr", 64, 20, U=16, seed=99)
    eng = api.Engine(64, info["samples_per_channel"], proto="dmr", ctx=gpu_ctx)
    eng.push(x); a = eng.frames(); sa = eng.symbols()
    eng.push(x); b = eng.frames()
    eng.reset()
    eng.push(x); c = eng.frames(); sc_ = eng.symbols()
    eng.close()
    B = 64
    assert (a[1] == c[1]).all() and (sa[1] == sc_[1]).all()
    assert _digest(a[0], a[1]) == _digest(c[0], c[1]) and _digest(sa[0], sa[1]) == _digest(sc_[0], sc_[1])
    assert int(a[1].sum()) > 0
    # the second push continued the stream (decoder already in sync, timing already settled): not the same bytes
    assert _digest(a[0], a[1]) != _digest(b[0], b[1])


def test_mixed_dmr_ysf_engines_on_two_streams(gpu_ctx, oracle):
    """BASELINE configs[4], one GPU's share: 8 192 DMR + 8 192 YSF channels as two engines created on their own HIP
    streams, pushed back to back (the launches may overlap on the device).  Replication over every channel plus the
    U distinct signals of each protocol against the oracle, two pushes (state carry)."""
    import torch
    from digiham_amd import api, synth_torch
    U, B = 32, 8192
    dev = gpu_ctx.mem.device
    parts = []
    for proto, units, seed in (("dmr", 33, 515), ("ysf", 12, 616)):
        base, info = synth_torch.make_batch(torch, dev, proto, U, units, U=U, seed=seed)
        stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(stream):                          # the engine captures the current stream
            eng = api.Engine(B, info["samples_per_channel"], proto=proto, ctx=gpu_ctx)
        parts.append({"proto": proto, "base": base, "x": base.repeat(B // U, 1).contiguous(), "eng": eng, "outs": []})
    torch.cuda.synchronize()
    for _ in range(2):
        for p in parts:                                          # both launches in flight before anything is read back
            p["eng"].push(p["x"])
        for p in parts:
            s, sc = p["eng"].symbols(); f, fc = p["eng"].frames(); e, ec = p["eng"].events()
            p["outs"].append((s, sc, f, fc, e, ec))
    for p in parts:
        p["eng"].close()
        for s, sc, f, fc, e, ec in p["outs"]:
            for rows, counts in ((s, sc), (f, fc), (e.view(np.uint8).reshape(B, -1), ec * 32)):
                d = _digest(rows, counts)
                assert all(d[ch] == d[ch % U] for ch in range(B))
        ref = oracle.chain(np.tile(p["base"].cpu().numpy(), (1, 2)), proto={"dmr": 1, "ysf": 2}[p["proto"]], threads=8)
        for b in range(U):
            gs = np.concatenate([o[0][b, :o[1][b]] for o in p["outs"]])
            gf = np.concatenate([o[2][b, :o[3][b]] for o in p["outs"]])
            ge = np.concatenate([o[4][b, :o[5][b]] for o in p["outs"]])
            assert len(gs) == ref["sym_count"][b] and (gs == ref["syms"][b, :len(gs)]).all()
            assert len(gf) == ref["out_count"][b] and (gf == ref["out"][b, :len(gf)]).all()
            assert ge.tobytes() == ref["events"][b, :ref["event_count"][b]].tobytes()
        assert sum(int(o[3].sum()) for o in p["outs"]) > 0


@pytest.mark.parametrize("proto", ["dmr", "ysf"])
def test_overlapped_pushes_equal_plain_pushes(gpu_ctx, proto):
    """DH_FLAG_OVERLAP_PUSHES: engines of >= 8192 DMR / YSF channels send a push as two launches on two streams of
    their own and join the caller's stream only when something is read.  Same bytes as plain pushes (ragged channel
    count, three pushes queued back to back before the first read), and the timing interface says what happened."""
    import torch
    from digiham_amd import api, synth_torch
    B, U = 8192 + 192, 64
    base, info = synth_torch.make_batch(torch, gpu_ctx.mem.device, proto, U, 16 if proto == "dmr" else 6, U=U, seed=777)
    T = info["samples_per_channel"]
    third = T // 3 // 10 * 10
    x = base.repeat(B // U, 1).contiguous()
    chunks = [x[:, :third].contiguous(), x[:, third:2 * third].contiguous(), x[:, 2 * third:].contiguous()]   # stay alive until the read
    got = []
    for overlap in (True, False):
        eng = api.Engine(B, T, proto=proto, ctx=gpu_ctx, overlap_pushes=overlap)
        eng.timing_enable(4)
        for c in chunks:
            eng.push(c)                                          # nothing read in between: with the flag the three pushes overlap
        eng.sync()
        outs = [tuple(a.copy() for pair in (eng.symbols(), eng.frames(), eng.events()) for a in pair)]      # of the last push
        first_ms, first_ch = eng.timing_read_split()
        eng.timing_read()
        assert len(first_ch) == 3
        if overlap:
            assert (first_ch == (B - B // 4) // 64 * 64).all() and (first_ms > 0).all()
        else:
            assert (first_ch == 0).all() and (first_ms == 0).all()
        eng.reset()                                              # joins the caller's stream, then overlapped again
        eng.push(x)
        outs.append(tuple(a.copy() for pair in (eng.symbols(), eng.frames(), eng.events()) for a in pair))
        eng.close()
        got.append(outs)
    for a, b in zip(*got):
        s, sc, f, fc, e, ec = a
        s2, sc2, f2, fc2, e2, ec2 = b
        assert (sc == sc2).all() and (fc == fc2).all() and (ec == ec2).all()
        for ch in range(B):
            assert (s[ch, :sc[ch]] == s2[ch, :sc[ch]]).all()
            assert (f[ch, :fc[ch]] == f2[ch, :fc[ch]]).all()
            assert e[ch, :ec[ch]].tobytes() == e2[ch, :ec[ch]].tobytes()
    assert int(got[0][1][3].sum()) > 0


@pytest.mark.parametrize("proto", ["dmr", "ysf"])
def test_error_bounded_and_exact_routes_agree_at_scale(gpu_ctx, proto):
    """The error-bounded FIR (default), the rounded-product FIR in every run (DH_FLAG_EXACT_FIR) and -- on a slice of the
    batch -- every symbol decided by exact arithmetic (DH_FLAG_EXACT_SYMBOLS) give the same dibits, frames and events
    on 16 384 channels of all three noise classes (a size-independent property: the three routes share no arithmetic
    beyond the staging)."""
    import torch
    from digiham_amd import api, synth_torch
    B = 16384
    x, info = synth_torch.make_batch(torch, gpu_ctx.mem.device, proto, B, 12 if proto == "dmr" else 4, U=192, seed=31337)
    T = info["samples_per_channel"]

    def run(n, **kw):
        eng = api.Engine(n, T, proto=proto, ctx=gpu_ctx, **kw)
        outs = []
        for _ in range(2):
            eng.push(x[:n])
            outs.append(tuple(a.copy() for pair in (eng.symbols(), eng.frames(), eng.events()) for a in pair))
        blocks, ordered = eng.timing_stats()
        eng.close()
        return outs, blocks, ordered

    ref, blocks, _ = run(B)
    assert int(blocks.sum()) > 0
    for n, kw in ((B, dict(exact_fir=True)), (768, dict(exact_symbols=True))):
        got, _, _ = run(n, **kw)
        for (s, sc, f, fc, e, ec), (s2, sc2, f2, fc2, e2, ec2) in zip(ref, got):
            assert (sc[:n] == sc2).all() and (fc[:n] == fc2).all() and (ec[:n] == ec2).all()
            for ch in range(n):
                assert (s[ch, :sc[ch]] == s2[ch, :sc[ch]]).all(), (kw, ch)
                assert (f[ch, :fc[ch]] == f2[ch, :fc[ch]]).all(), (kw, ch)
                assert e[ch, :ec[ch]].tobytes() == e2[ch, :ec[ch]].tobytes(), (kw, ch)


def test_overlapped_pushes_keep_their_temporary_inputs_alive(gpu_ctx):
    """With DH_FLAG_OVERLAP_PUSHES the kernels read a pushed buffer on the engine's own streams, later than the caller's
    stream knows: Engine.push holds every input until the streams are joined.  Pushes of temporaries, with the allocator
    handing the same blocks out again for garbage in between, must give the bytes of plain pushes."""
    import torch
    from digiham_amd import api, synth_torch
    B, U = 8192, 64
    base, info = synth_torch.make_batch(torch, gpu_ctx.mem.device, "dmr", U, 12, U=U, seed=31)
    T = info["samples_per_channel"]
    x = base.repeat(B // U, 1).contiguous()
    cut = [0, T // 4 // 10 * 10, T // 2 // 10 * 10, 3 * T // 4 // 10 * 10, T]
    got = []
    for overlap in (True, False):
        eng = api.Engine(B, T, proto="dmr", ctx=gpu_ctx, overlap_pushes=overlap)
        sym_total = np.zeros(B, np.int64)
        for a, b in zip(cut[:-1], cut[1:]):
            eng.push(x[:, a:b].contiguous())                     # a temporary: nothing here keeps it alive
            for _ in range(3):                                   # the allocator is invited to reuse freed blocks of this size at once
                junk = torch.full((B, b - a), float("nan"), device=x.device)
                del junk
        s, sc = eng.symbols()
        f, fc = eng.frames()
        got.append((_digest(s, sc), _digest(f, fc)))
        eng.close()
    assert got[0] == got[1]


@pytest.mark.parametrize("fast", [False, True, "one_launch"])
def test_config1_rrc_materialised_plus_gfsk_at_full_size(gpu_ctx, oracle, fast):
    """BASELINE configs[1] as stated: 4 096 channels, RRC output materialised (k_rrc_tile's 2-D grid + k_rrc_hist) + GFSK
    slicer, two pushes.  Replication digest over ALL channels (32 distinct signals repeated), the 32 against the oracle:
    floats bit-exact (exact FIR) or within 1e-6 relative to max(|ref|, rms) (FMA FIR), dibits bit-exact (exact FIR -- and the
    one-launch FMA mode, DH_FLAG_ONE_LAUNCH | DH_FLAG_FAST_FIR, whose slicer is the error-bounded one)."""
    import torch
    from digiham_amd import api, synth_torch
    B, U = 4096, 32
    base, info = synth_torch.make_batch(torch, gpu_ctx.mem.device, "dmr", U, 33, U=U, seed=555)
    T = info["samples_per_channel"]
    x = base.repeat(B // U, 1).contiguous()
    eng = api.Engine(B, T, proto="none", keep_filtered=True, fast_fir=bool(fast), one_launch=fast == "one_launch", ctx=gpu_ctx)
    outs = []
    for _ in range(2):
        eng.push(x)
        s, sc = eng.symbols()
        y = eng.filtered()[:, :T].copy()
        outs.append((s, sc, y))
    eng.close()
    for s, sc, y in outs:
        d = _digest(s, sc)
        dy = [hashlib.sha256(y[b].tobytes()).hexdigest() for b in range(B)]
        # (the one-launch FMA mode: the last runs of the LAST channel -- their window would reach beyond the input buffer -- take the reference-order
        # FIR, so that row carries the reference's own floats there: compared with the oracle below instead of with its replica)
        rows = range(B - 1) if fast == "one_launch" else range(B)
        assert all(d[ch] == d[ch % U] for ch in range(B)) and all(dy[ch] == dy[ch % U] for ch in rows)
    ref = oracle.chain(np.tile(base.cpu().numpy(), (1, 2)), proto=0, keep_filtered=True, threads=8)
    if fast == "one_launch":
        last = np.concatenate([o[2][B - 1] for o in outs]); rl = ref["filtered"][(B - 1) % U]
        assert float(np.max(np.abs(last.astype(np.float64) - rl) / np.maximum(np.abs(rl), np.sqrt(np.mean(rl.astype(np.float64) ** 2))))) <= 1e-6
    yy = np.concatenate([o[2][:U] for o in outs], axis=1)
    r = ref["filtered"]
    if fast:
        rms = np.sqrt(np.mean(r.astype(np.float64) ** 2)) + 1e-30
        assert float(np.max(np.abs(yy.astype(np.float64) - r) / np.maximum(np.abs(r), rms))) <= 1e-6
    else:
        assert (yy.view(np.uint32) == r.view(np.uint32)).all()
    if fast is not True:
        for b in range(U):
            gs = np.concatenate([o[0][b, :o[1][b]] for o in outs])
            assert len(gs) == ref["sym_count"][b] and (gs == ref["syms"][b, :len(gs)]).all()


def test_bench_runs_its_distributed_path_on_one_gpu():
    """bench.py with DH_FORCE_DIST=1: the RCCL process group is initialised on the single rank, so init / barrier /
    all_reduce of the multi-GPU path execute on the GPU box (the 8-GPU run itself is the driver's)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--channels", "2048", "--units", "20", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-other-configs", "--verify", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["verified"]["bit_exact_vs_oracle"]
    # the line the driver parses is the LAST one of stdout, at most 4 KB, with the roofline in it; the full record comes before it
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert last == lines[0] and len(last) <= 4096 and 0 < line["roofline"]["frac"] < 1 and line["roofline"]["avg_launch_ms"] > 0
    detail = [l for l in r.stdout.splitlines() if l.startswith("BENCH_DETAIL ")]
    assert len(detail) == 1 and json.loads(detail[0][len("BENCH_DETAIL "):])["roofline"]["peak_achievable_rates"]


@pytest.mark.gpu
def test_bench_iq_front_end_on_its_own_stream_equals_the_serial_order(gpu_ctx):
    """bench.py `--workload dmr_iq_full --streams 2`: the front-end of step k + 1 runs on its own stream beside the chain kernel of
    step k (two float buffers, events for "converted before pushed" and "pushed before overwritten").  Same symbols, frames and
    events behind bursts of 1, 4 and 3 queued steps as with both on one stream."""
    import torch
    import bench
    dev = gpu_ctx.mem.device
    outs = []
    for streams in (1, 2):
        job = bench.Job(torch, gpu_ctx, dev, "dmr_iq_full", 512, rank=0, streams=streams, units=40)
        got = []
        for burst in (1, 4, 3):                               # steps queued back to back (the streams really overlap), outputs read behind each burst
            for _ in range(burst):
                job.step()
            e = job.parts[0]["eng"]
            s, sc = e.symbols(); f, fc = e.frames(); ev, ec = e.events()
            got.append(_digest(s, sc) + _digest(f, fc) + [hashlib.sha256(ev[b, :ec[b]].tobytes()).hexdigest() for b in range(len(ec))])
        job.close()
        outs.append(got)
    assert outs[0] == outs[1]
    assert len(set(outs[0][-1])) > 1
