"""The Csdr::Module-shaped C++ classes of include/digiham/ (same names and constructor signatures as the
reference's include/*.hpp), driven like the reference CLI drives its modules, against the oracle.

CPU tier: linked against the wave-emulation library.  GPU tier (-m gpu): linked against libdigiham_amd.so.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from digiham_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_pipe_test(tmp_path, gpu):
    exe = str(tmp_path / ("pipe_test_gpu" if gpu else "pipe_test_emu"))
    if gpu:
        libdir, lib = os.path.join(ROOT, "digiham_amd"), "digiham_amd"
    else:
        import hostemu
        hostemu.build()
        libdir, lib = os.path.join(ROOT, "tests", "host_harness"), "dh_hostemu"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host_cpp", "pipe_test.cpp"), "-o", exe,
                    "-L" + libdir, "-l" + lib, "-Wl,-rpath," + libdir], check=True)
    return exe


def run_pipe(exe, proto, x, tmp_path, chunk, gpu):
    inp = tmp_path / "in.f32"
    x.astype(np.float32).tofile(inp)
    prefix = str(tmp_path / ("o_%s_%d" % (proto, chunk)))
    env = dict(os.environ)
    if gpu:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    subprocess.run([exe, proto, str(inp), prefix, str(chunk)], check=True, env=env)
    rd = lambda suffix, dt: np.fromfile(prefix + suffix, dt)
    return {"filtered": rd(".filtered", np.float32), "syms": rd(".syms", np.uint8), "out": rd(".out", np.uint8),
            "events": rd(".events", api.EVENT_DTYPE), "dvin": rd(".dvin", np.int16), "dvout": rd(".dvout", np.int16),
            "meta": rd(".meta", np.uint8).tobytes(), "smallmeta": rd(".smallmeta", np.uint8).tobytes()}


@pytest.mark.parametrize("gpu", [False, pytest.param(True, marks=pytest.mark.gpu)])
@pytest.mark.parametrize("proto", ["dmr", "ysf"])
def test_module_classes_match_oracle(oracle, tmp_path, proto, gpu):
    exe = build_pipe_test(tmp_path, gpu)
    s = synth.dmr_stream(61, 24) if proto == "dmr" else synth.ysf_stream(62, 8)
    x = synth.impair(synth.shape(s), 61, snr_db=22, dc=0.05, delay=5)
    ref = oracle.chain(x[None, :], proto=1 if proto == "dmr" else 2, keep_filtered=True)
    for chunk in (4096, 777):
        got = run_pipe(exe, proto, x, tmp_path, chunk, gpu)
        assert got["filtered"].tobytes() == ref["filtered"][0].tobytes()
        # the module consumes its reader completely; the oracle (like the reference) leaves the last <= sps+1
        # samples unread, so the module may have produced at most one more symbol than the reference pipe would
        ns = int(ref["sym_count"][0])
        assert len(got["syms"]) in (ns, ns + 1) and (got["syms"][:ns] == ref["syms"][0, :ns]).all()
        d = oracle.Decoder(proto)
        o, ev = d.process(got["syms"])
        assert (got["out"] == o).all()
        assert got["events"].tobytes() == ev.tobytes()
        assert (got["dvout"] == oracle.DvFilter().process(got["dvin"])).all()
        # PipelineMetaWriter: whole `k:v;k:v\n` lines of this protocol in the pipeline buffer; too-long lines dropped whole
        lines = got["meta"].split(b"\n")
        assert lines[-1] == b"" and len(lines) > 2
        assert all((b"protocol:%s" % proto.upper().encode()) in l.split(b";") for l in lines[:-1])
        if proto == "dmr":
            lcs = ev[ev["type"] == 4]
            f = api.parse_lc(bytes(lcs[0]["payload"][:9]))
            assert any(b"source:%d;" % f["source"] in l and b"target:%d;type:group" % f["target"] in l for l in lines)
        else:
            assert b"mode:DN;protocol:YSF" in got["meta"]
        assert got["smallmeta"] == b"a:b\n"


@pytest.mark.parametrize("gpu", [False, pytest.param(True, marks=pytest.mark.gpu)])
def test_rrc_filter_with_a_foreign_tap_table(oracle, tmp_path, gpu):
    """Digiham::RrcFilter::RrcFilter(nZeros, gain, coeffs[]) (include/rrc_filter.hpp:12) with a table that is neither of
    the built-in designs: 41 random, non-symmetric coefficients -- bit-exact against the oracle's FIR with the same table."""
    rng = np.random.default_rng(5)
    taps = rng.normal(0, 1, 41).astype(np.float32)
    gain = 3.217
    x = (rng.normal(0, 1, 20000) * 10.0 ** rng.uniform(-3, 1, 20000)).astype(np.float32)
    src = tmp_path / "t.cpp"
    src.write_text("""
#include <cstdio>
#include <vector>
#include "digiham/rrc_filter.hpp"
int main() {
    std::vector<float> taps(41), x(20000), y(20000);
    if (fread(taps.data(), 4, 41, stdin) != 41 || fread(x.data(), 4, x.size(), stdin) != x.size()) return 2;
    try {
        Digiham::RrcFilter::RrcFilter f(40, %r, taps.data());
        struct R: Csdr::Reader<float> { std::vector<float>* v; size_t pos = 0, lim = 0; size_t available() override { return lim - pos; }
            float* getReadPointer() override { return v->data() + pos; } void advance(size_t n) override { pos += n; } } r;
        struct W: Csdr::Writer<float> { std::vector<float>* v; size_t pos = 0; size_t writeable() override { return v->size() - pos; }
            float* getWritePointer() override { return v->data() + pos; } void advance(size_t n) override { pos += n; } } w;
        r.v = &x; w.v = &y;
        Csdr::Module<float, float>* m = &f;
        m->setReader(&r); m->setWriter(&w);
        for (size_t step : { 3000, 17, 5000, 11983 }) { r.lim += step; while (m->canProcess()) m->process(); }
        fwrite(y.data(), 4, w.pos, stdout);
    } catch (const std::exception& e) { fprintf(stderr, "%%s\\n", e.what()); return 1; }
    return 0;
}
""" % gain)
    if gpu:
        libdir, lib = os.path.join(ROOT, "digiham_amd"), "digiham_amd"
    else:
        import hostemu
        hostemu.build()
        libdir, lib = os.path.join(ROOT, "tests", "host_harness"), "dh_hostemu"
    exe = str(tmp_path / "t")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe, "-L" + libdir, "-l" + lib,
                    "-Wl,-rpath," + libdir], check=True)
    env = dict(os.environ)
    if gpu:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], input=taps.tobytes() + x.tobytes(), capture_output=True, check=True, env=env).stdout
    got = np.frombuffer(out, np.float32)
    ref = oracle.Rrc(taps=taps, gain=gain).process(x)
    assert len(got) == len(x) and got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("gpu", [False, pytest.param(True, marks=pytest.mark.gpu)])
def test_shared_bank_mixes_decoders_with_and_without_a_meta_writer(oracle, tmp_path, gpu):
    """A decoder bank whose instances do not all have a meta writer: events are fetched for the slots that asked for them only,
    and an instance without a consumer never sees "undelivered output" that nobody will take -- the driver's
    `while canProcess(): process()` terminates (shared_test exits 4 on a livelock), every channel's bytes are the oracle's,
    and exactly the instances with a writer have metadata lines."""
    N = 12
    exe = str(tmp_path / ("mixed_gpu" if gpu else "mixed_emu"))
    if gpu:
        libdir, lib = os.path.join(ROOT, "digiham_amd"), "digiham_amd"
    else:
        import hostemu
        hostemu.build()
        libdir, lib = os.path.join(ROOT, "tests", "host_harness"), "dh_hostemu"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_cpp", "shared_test.cpp"),
                    "-o", exe, "-L" + libdir, "-l" + lib, "-Wl,-rpath," + libdir], check=True)
    chans = []
    for i in range(N):
        s = synth.dmr_stream(300 + i % 4, 14, two_slots=(i % 2 == 0))
        chans.append(synth.impair(synth.shape(s), 300 + i, snr_db=[None, 25, 16][i % 3], dc=0.02 * (i % 5), delay=i % 17, gain=[1, 0.4, 2][i % 3]))
    T = min(len(c) for c in chans)
    x = np.stack([c[:T] for c in chans]).astype(np.float32)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    env = dict(os.environ)
    if gpu:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    prefix = str(tmp_path / "m")
    r = subprocess.run([exe, str(N), str(tmp_path / "in.f32"), str(T), prefix, "3000", "1", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr)
    rounds, ticks = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    assert 0 < ticks <= 3 * rounds
    outs = [open("%s.%d.out" % (prefix, i), "rb").read() for i in range(N)]
    metas = [open("%s.%d.meta" % (prefix, i), "rb").read() for i in range(N)]
    for i in range(N):
        ref = oracle.chain(x[i:i + 1], proto=1, slot_filter=1 if i % 5 == 4 else 3)
        assert outs[i] == ref["out"][0, :ref["out_count"][0]].tobytes(), i
        assert (b"protocol:DMR" in metas[i]) == (i % 3 == 0), i
    assert sum(len(o) for o in outs) > 0


@pytest.mark.parametrize("gpu", [False, pytest.param(True, marks=pytest.mark.gpu)])
def test_64_module_triples_share_one_launch_per_stage_and_round(oracle, tmp_path, gpu):
    """Digiham::Amd::SharedEngine (include/digiham/shared_engine.hpp): 64 x (WideRrcFilter | GfskDemodulator | Dmr::Decoder) in
    one process, fed raggedly, driven round-robin.  Every channel's decoder bytes and metadata lines are those of its own
    1-channel engines (and the bytes the oracle's), with at most one launch per stage and round instead of one per module."""
    N = 64
    exe = str(tmp_path / ("shared_gpu" if gpu else "shared_emu"))
    if gpu:
        libdir, lib = os.path.join(ROOT, "digiham_amd"), "digiham_amd"
    else:
        import hostemu
        hostemu.build()
        libdir, lib = os.path.join(ROOT, "tests", "host_harness"), "dh_hostemu"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_cpp", "shared_test.cpp"),
                    "-o", exe, "-L" + libdir, "-l" + lib, "-Wl,-rpath," + libdir], check=True)
    chans = []
    for i in range(N):
        s = synth.dmr_stream(100 + i % 8, 14, two_slots=(i % 2 == 0))
        chans.append(synth.impair(synth.shape(s), 100 + i, snr_db=[None, 25, 16][i % 3], dc=0.02 * (i % 5), delay=i % 17, gain=[1, 0.4, 2][i % 3]))
    T = min(len(c) for c in chans)
    x = np.stack([c[:T] for c in chans]).astype(np.float32)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    env = dict(os.environ)
    if gpu:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    res = {}
    for shared in (1, 0):
        prefix = str(tmp_path / ("s%d" % shared))
        r = subprocess.run([exe, str(N), str(tmp_path / "in.f32"), str(T), prefix, "3000", str(shared)], check=True, env=env, capture_output=True, text=True)
        rounds, ticks = int(r.stdout.split()[1]), int(r.stdout.split()[3])
        res[shared] = ([open("%s.%d.out" % (prefix, i), "rb").read() for i in range(N)],
                       [open("%s.%d.meta" % (prefix, i), "rb").read() for i in range(N)], rounds, ticks)
    outs, metas, rounds, ticks = res[1]
    assert outs == res[0][0] and metas == res[0][1]                 # the same bytes and lines as with one engine per module
    assert res[0][3] == 0 and 0 < ticks <= 3 * rounds               # at most one launch per stage and round (192 per round without)
    for i in range(0, N, 7):                                        # and the bytes are the oracle's
        ref = oracle.chain(x[i:i + 1], proto=1, slot_filter=1 if i % 5 == 4 else 3)
        assert outs[i] == ref["out"][0, :ref["out_count"][0]].tobytes()
    assert sum(len(o) for o in outs) > 0 and all(b"protocol:DMR" in m for m in metas)
