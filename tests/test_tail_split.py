"""The tail split of the chain launches (engine.hip: k_chain, HipBackend::go_chain).

A launch of one workgroup per channel drains for about one workgroup's duration; the chain kernels therefore hand the
last part of every row to a second workgroup of the same launch, which starts from the state the first one wrote back and
appends its symbols, frames and events.  A part is a push, so nothing may change: these tests compare split launches with
whole ones and with the oracle -- on the CPU tier through the wave emulation (tests/host_harness/harness.cpp runs the two
parts of a channel one after the other: the part arithmetic of the kernel bodies), on the GPU at a size where the
engine really splits (>= 8192 channels, >= 65536 samples per push), with a channel count that is not a multiple of 8
(the grid is padded), ragged counts on both sides of the split point and a short push (not split) in between.
"""
import hashlib

import numpy as np
import pytest

from common import make_channels
from digiham_amd import api


def _collect(eng, B, acc):
    for k, (rows, counts) in enumerate((eng.symbols(), eng.frames(), eng.events())):
        for b in range(B):
            acc[k][b].append(rows[b, :counts[b]].copy())


@pytest.mark.parametrize("proto,oproto", [("dmr", 1), ("ysf", 2)])
@pytest.mark.parametrize("pct", ["1", "37", "75", "99", "40,90", "80,81", "75:fail", "40,90:fail"])
def test_split_pushes_on_the_wave_emulation(emu_ctx, oracle, monkeypatch, proto, oproto, pct):
    x = make_channels(proto, [31, 32, 33, 34], 10)
    B, n = x.shape
    ref = oracle.chain(x, proto=oproto)
    if pct.endswith(":fail"):                   # channels 1 and 3 lose their hand-over: the rest of the row in one piece (the fix-up launch's arithmetic)
        pct = pct[:-5]
        monkeypatch.setenv("DH_TAIL_SPLIT_FORCE_FAIL", "2")
    monkeypatch.setenv("DH_TAIL_SPLIT", str(pct))
    cap = 20000
    eng = api.Engine(B, cap, proto=proto, ctx=emu_ctx)
    rng = np.random.default_rng(len(pct) + int(pct.split(",")[0]))
    acc = [[[] for _ in range(B)] for _ in range(3)]
    pos = np.zeros(B, np.int64)
    while (pos < n).any():
        # ragged counts: some channels end in front of the split point (their second part is empty), some bring nothing
        want = rng.choice([0, 3, 150, 5000, 12345, cap], B)
        cnt = np.minimum(want, n - pos).astype(np.uint32)
        buf = np.full((B, cap), np.nan, np.float32)
        for b in range(B):
            buf[b, :cnt[b]] = x[b, pos[b]:pos[b] + cnt[b]]
        eng.push(buf, n=int(cnt.max()), counts=cnt)
        pos += cnt
        _collect(eng, B, acc)
    eng.close()
    for b in range(B):
        gs, gf, ge = (np.concatenate(acc[k][b]) for k in range(3))
        assert len(gs) == ref["sym_count"][b] and (gs == ref["syms"][b, :len(gs)]).all(), b
        assert len(gf) == ref["out_count"][b] and (gf == ref["out"][b, :len(gf)]).all(), b
        assert ge.tobytes() == ref["events"][b, :ref["event_count"][b]].tobytes(), b
    assert sum(len(np.concatenate(acc[1][b])) for b in range(B)) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("proto,oproto,kw,okw", [("dmr", 1, {}, {}), ("ysf", 2, {}, {}),
                                                ("nxdn", 3, dict(rrc="narrow", sps=20), dict(rrc=2, sps=20))])
def test_split_launch_equals_whole_launch_and_oracle(gpu_ctx, oracle, monkeypatch, proto, oproto, kw, okw):
    import torch
    from digiham_amd import synth_torch
    U, B = 16, 8192 + 35                                        # not a multiple of 8: the grid of a split launch is padded
    units = {"dmr": 50, "ysf": 15, "nxdn": 19}[proto]
    base, info = synth_torch.make_batch(torch, gpu_ctx.mem.device, proto, U, units, U=U, seed=777, sps=kw.get("sps", 10))
    T = info["samples_per_channel"]
    assert T - 5000 >= 65536                                    # (DH_TAIL_SPLIT_MIN_SAMPLES)
    reps = (B + U - 1) // U
    x = base.repeat(reps, 1)[:B].contiguous()
    n1, n2 = T - 5000, 5000                                     # a long push (split), then a short one (not split)
    rng = np.random.default_rng(3)
    # ragged first push: signal u brings cnt[u] samples -- in front of, at and behind the split point of every setting
    cnt_u = np.array([n1, n1, 0, 1, n1 * 3 // 4 - 1, n1 * 3 // 4, n1 * 3 // 4 + 1, n1 // 2, n1 - 777, n1 - 1, n1, 12345, n1, n1 * 9 // 10, n1, n1], np.uint32)
    counts = np.tile(cnt_u, reps)[:B].copy()
    results = {}
    for pct in ("0", "80", "50", "70,92", "80:fail", "70,92:fail"):
        # ":fail" -- the later parts of every third channel give up as if their hand-over had not come (DH_TAIL_SPLIT_FORCE_FAIL):
        # the fix-up launch behind the split launch finishes those rows, and nothing is reported
        monkeypatch.setenv("DH_TAIL_SPLIT", pct.split(":")[0])
        if pct.endswith(":fail"):
            monkeypatch.setenv("DH_TAIL_SPLIT_FORCE_FAIL", "3")
        else:
            monkeypatch.delenv("DH_TAIL_SPLIT_FORCE_FAIL", raising=False)
        eng = api.Engine(B, T, proto=proto, ctx=gpu_ctx, **kw)
        acc = [[[] for _ in range(B)] for _ in range(3)]
        eng.push(x, n=n1, counts=torch.from_numpy(counts).to(x.device))
        _collect(eng, B, acc)
        # every channel continues where it stopped: the rest of its row in a second (whole) and third (short) push
        rest = torch.zeros_like(x)
        cnt2 = (T - counts).astype(np.uint32)
        for u in range(U):
            rest[u::U, :T - int(cnt_u[u])] = x[u::U, int(cnt_u[u]):]
        first = np.where(cnt2 > n2, cnt2 - n2, 0).astype(np.uint32)
        eng.push(rest, n=int(first.max()), counts=torch.from_numpy(first).to(x.device))
        _collect(eng, B, acc)
        tail = torch.zeros((B, n2), dtype=x.dtype, device=x.device)
        last = (cnt2 - first).astype(np.uint32)
        for u in range(U):
            f0, l0 = int(first[u]), int(last[u])
            tail[u::U, :l0] = rest[u::U, f0:f0 + l0]
        eng.push(tail, n=int(last.max()), counts=torch.from_numpy(last).to(x.device))
        _collect(eng, B, acc)
        gave_up = int(eng.debug_header(202)[0])                 # hand-overs that did not come (finished by the fix-up launch)
        assert (gave_up > 0) == pct.endswith(":fail"), (pct, gave_up)
        if pct == "70,92:fail":
            # three parts: the forced failure is the SECOND part's; the third must leave because it is told so (the give-up bit in a word
            # of this push), not because its patience (2^16 naps, ~130 ms per channel) ran out -- every one of them
            told = int(eng.debug_header(203)[0])
            assert told > 0 and 2 * told == gave_up, (told, gave_up)
        eng.close()
        results[pct] = [[hashlib.sha256(np.concatenate(acc[k][b]).tobytes()).hexdigest() for b in range(B)] for k in range(3)]
        if pct == "0":
            whole = [[np.concatenate(acc[k][b]) for b in range(U)] for k in range(3)]
    for pct in ("80", "50", "70,92", "80:fail", "70,92:fail"):
        assert results[pct] == results["0"], pct
    assert all(results["0"][k][b] == results["0"][k][b % U] for k in range(3) for b in range(B))
    ref = oracle.chain(base.cpu().numpy(), proto=oproto, threads=8, **okw)
    for b in range(U):
        gs, gf, ge = whole[0][b], whole[1][b], whole[2][b]
        assert len(gs) == ref["sym_count"][b] and (gs == ref["syms"][b, :len(gs)]).all(), b
        assert len(gf) == ref["out_count"][b] and (gf == ref["out"][b, :len(gf)]).all(), b
        assert ge.tobytes() == ref["events"][b, :ref["event_count"][b]].tobytes(), b
    assert sum(len(whole[1][b]) for b in range(U)) > 0


@pytest.mark.gpu
def test_two_split_engines_on_two_streams(gpu_ctx, oracle, monkeypatch):
    """What `bench.py --workload mixed --streams 2` times: an 8 192-channel DMR and an 8 192-channel YSF engine on their own HIP
    streams, pushes long enough for the tail split (>= 65 536 samples) on BOTH, three pushes queued back to back before
    anything is read.  Same bytes as the unsplit engines, every channel; the distinct signals against the oracle."""
    import torch
    from digiham_amd import synth_torch
    U, B = 32, 8192
    dev = gpu_ctx.mem.device
    hashes = {}
    for pct in ("0", "80"):
        monkeypatch.setenv("DH_TAIL_SPLIT", pct)
        parts = []
        for proto, units, seed in (("dmr", 50, 515), ("ysf", 15, 616)):
            base, info = synth_torch.make_batch(torch, dev, proto, U, units, U=U, seed=seed)
            assert info["samples_per_channel"] >= 65536
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):                      # the engine captures the current stream
                eng = api.Engine(B, info["samples_per_channel"], proto=proto, ctx=gpu_ctx)
            parts.append({"proto": proto, "base": base, "x": base.repeat(B // U, 1).contiguous(), "eng": eng, "outs": []})
        torch.cuda.synchronize()
        for _ in range(2):
            for p in parts:
                p["eng"].push(p["x"])
            for p in parts:
                p["eng"].push(p["x"])                           # a second push of each queued behind the first, both streams busy
            for p in parts:
                p["outs"].append(tuple(a.copy() for pair in (p["eng"].symbols(), p["eng"].frames(), p["eng"].events()) for a in pair))
        for p in parts:
            assert int(p["eng"].debug_header(202)[0]) == 0      # every hand-over came
            p["eng"].close()
            hashes[(pct, p["proto"])] = [[hashlib.sha256(o[2 * k][b, :o[2 * k + 1][b]].tobytes()).hexdigest() for b in range(B)] for o in p["outs"] for k in range(3)]
            if pct == "80":
                # pushes 2 and 4 of the stream were read: compare them with the oracle's stream of four
                ref = oracle.chain(np.tile(p["base"].cpu().numpy(), (1, 4)), proto={"dmr": 1, "ysf": 2}[p["proto"]], threads=8)
                pre = [oracle.chain(np.tile(p["base"].cpu().numpy(), (1, k)), proto={"dmr": 1, "ysf": 2}[p["proto"]], threads=8) for k in (1, 2, 3)]
                for b in range(U):
                    for o, lo, hi in ((p["outs"][0], pre[0], pre[1]), (p["outs"][1], pre[2], ref)):
                        for k, (rows, cnt) in enumerate((("syms", "sym_count"), ("out", "out_count"), ("events", "event_count"))):
                            got = o[2 * k][b, :o[2 * k + 1][b]]
                            want = hi[rows][b, lo[cnt][b]:hi[cnt][b]]
                            assert got.tobytes() == want.tobytes(), (p["proto"], b, rows)
                assert sum(int(o[3].sum()) for o in p["outs"]) > 0
    for proto in ("dmr", "ysf"):
        assert hashes[("80", proto)] == hashes[("0", proto)], proto
