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


def test_rrc_rejects_foreign_tap_tables(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "digiham/rrc_filter.hpp"\nint main(){ float c[3]={1,2,1}; try { Digiham::RrcFilter::RrcFilter f(2, 4.0, c); } '
                   'catch (const std::invalid_argument&) { return 0; } return 1; }\n')
    import hostemu
    hostemu.build()
    libdir = os.path.join(ROOT, "tests", "host_harness")
    exe = str(tmp_path / "t")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe, "-L" + libdir, "-ldh_hostemu",
                    "-Wl,-rpath," + libdir], check=True)
    assert subprocess.run([exe]).returncode == 0
