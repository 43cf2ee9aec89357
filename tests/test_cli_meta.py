"""the reference's command line tools and metadata lines on the engine.

* metadata collectors (include/digiham/{meta,dmr_meta,ysf_meta}.hpp) replaying decoder events into the reference's
  `k:v;k:v` lines -- hand-derived expectations from src/lib/meta.cpp:8-17, dmr_meta.cpp:91-111, ysf_meta.cpp:13-45
  (PARITY UNPINNED: the reference has no tests and cannot be built here);
* the tools of cli/ driven as in examples/dmr-decoder.sh (`rrc_filter | gfsk_demodulator | dmr_decoder --fifo`),
  stdout against the oracle; CPU tier = linked against the wave emulation, GPU tier = cli/bin on the MI355X.
"""
import os
import subprocess

import numpy as np
import pytest

from digiham_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = ["rrc_filter", "gfsk_demodulator", "fsk_demodulator", "digitalvoice_filter", "dmr_decoder", "ysf_decoder", "nxdn_decoder", "pocsag_decoder", "dstar_decoder"]


def _ev(type_, a=0, b=0, payload=b""):
    e = np.zeros(1, api.EVENT_DTYPE)
    e["type"], e["a"], e["b"], e["len"] = type_, a, b, len(payload)
    e["payload"][0, :len(payload)] = np.frombuffer(payload, np.uint8)
    return e


def _batches(batches):
    out = b""
    for evs in batches:
        out += np.uint32(len(evs)).tobytes() + b"".join(e.tobytes() for e in evs)
    return out


def _meta_test(tmp_path):
    exe = str(tmp_path / "meta_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host_cpp", "meta_test.cpp"), "-o", exe], check=True)
    return exe


def test_dmr_meta_lines(tmp_path):
    exe = _meta_test(tmp_path)
    lc = synth.dmr_lc(0, 0, 0, 1234, 5678901)
    alias = bytes([4, 0, 0x40 | (6 << 1)]) + b"DL1ABC"                      # talker alias header, 8-bit format, length 6
    gps = bytes([8, 0, 0x00, 0x10, 0x00, 0x00, 0x20, 0x00, 0x00])           # lon 360/32, lat 180/8
    evs = [[_ev(1, 0, 2, b"\x00"), _ev(4, 0, 1, lc), _ev(4, 0, 1, lc)],
           [_ev(4, 0, 1, alias), _ev(4, 0, 1, gps)],
           [_ev(1, 0, 1, b"\x01"), _ev(2, 0), _ev(3)]]
    got = subprocess.run([exe, "dmr"], input=_batches(evs), capture_output=True, check=True).stdout.decode().splitlines()
    assert got == [
        "protocol:DMR;slot:0;sync:voice",
        "protocol:DMR;slot:0;source:5678901;sync:voice;target:1234;type:group",
        "protocol:DMR;slot:0;source:5678901;sync:voice;talkeralias:DL1ABC;target:1234;type:group",
        "lat:22.500000;lon:11.250000;protocol:DMR;slot:0;source:5678901;sync:voice;talkeralias:DL1ABC;target:1234;type:group",
        "protocol:DMR;slot:0;sync:data",
        "protocol:DMR;slot:0",
    ]


def test_ysf_meta_lines(tmp_path):
    exe = _meta_test(tmp_path)
    evs = [[_ev(17, 0, 2), _ev(18, 0, 0, b"CQCQCQ    "), _ev(18, 1, 0, b"DL1ABC    ")],
           [_ev(20, 0, 1), _ev(19, 0, 0, b"ALL       DL1ABC    "), _ev(19, 1, 0, b"DB0XYZ    DB0XYZ    ")],
           [_ev(20, 0, 2)]]
    got = subprocess.run([exe, "ysf"], input=_batches(evs), capture_output=True, check=True).stdout.decode().splitlines()
    assert got == [
        "mode:DN;protocol:YSF",
        "mode:DN;protocol:YSF;target:CQCQCQ",
        "mode:DN;protocol:YSF;source:DL1ABC;target:CQCQCQ",
        "protocol:YSF",
        "down:DB0XYZ;protocol:YSF;source:DL1ABC;target:ALL;up:DB0XYZ",
        "protocol:YSF",
    ]


def test_ysf_gps_from_the_data_frames(tmp_path):
    """V/D2 data frames 6 and 7 carry DT1 / DT2; a short-GPS command gives lat / lon (data.cpp:72-87, gps.cpp:5-82)."""
    exe = _meta_test(tmp_path)
    dt = bytearray(20)
    dt[1:4] = bytes([0x22, 0x62, 0x5F])                                   # COMMAND_SHORT_GPS
    dt[4] = 0x28
    dt[5:14] = bytes([0x34, 0x38, 0x31, 0x52, 0x33, 0x34, 0x27, 0x30, 0x4E])
    dt[18] = 0x03
    dt[19] = sum(dt[:19]) & 0xFF
    f = np.float32
    lat = f(f(f(f(f(48) + f(1) / f(6)) + f(2) / f(60)) + f(3) / f(600)) + f(4) / f(6000))
    lon = f(f(f(11) + f(20) / f(60)) + f(50) / f(6000))
    evs = [[_ev(17, 0, 2), _ev(18, 6, 0, bytes(dt[:10])), _ev(18, 7, 0, bytes(dt[10:]))], [_ev(18, 0, 0, b"CQCQCQ    ")], [_ev(20, 0, 2)]]
    got = subprocess.run([exe, "ysf"], input=_batches(evs), capture_output=True, check=True).stdout.decode().splitlines()
    assert got == [
        "mode:DN;protocol:YSF",
        "lat:%f;lon:%f;mode:DN;protocol:YSF" % (lat, lon),
        "lat:%f;lon:%f;mode:DN;protocol:YSF;target:CQCQCQ" % (lat, lon),
        "protocol:YSF",
    ]


def test_nxdn_meta_lines(tmp_path):
    exe = _meta_test(tmp_path)
    vcall = bytes([0x01, 0x00, 0x20, 0x12, 0x34, 0x00, 0x63, 0, 0])         # conference call 0x1234 -> 99
    evs = [[_ev(35), _ev(35), _ev(34, 0, 0, vcall)], [_ev(34, 0, 0, vcall), _ev(37, 0, 1)]]
    got = subprocess.run([exe, "nxdn"], input=_batches(evs), capture_output=True, check=True).stdout.decode().splitlines()
    assert got == [
        "protocol:NXDN;sync:voice",
        "protocol:NXDN;sync:voice;type:conference",
        "protocol:NXDN;source:4660;sync:voice;type:conference",
        "destination:99;protocol:NXDN;source:4660;sync:voice;type:conference",
        "protocol:NXDN",
    ]


def _build_tools(tmp_path, gpu):
    if gpu:
        subprocess.run(["make", "-C", os.path.join(ROOT, "cli"), "-s"], check=True)
        return os.path.join(ROOT, "cli", "bin")
    import hostemu
    hostemu.build()
    libdir = os.path.join(ROOT, "tests", "host_harness")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    for t in TOOLS:
        subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "cli", t + ".cpp"),
                        "-o", str(bindir / t), "-L" + libdir, "-ldh_hostemu", "-Wl,-rpath," + libdir, "-pthread"], check=True)
    return str(bindir)


@pytest.mark.parametrize("gpu", [False, pytest.param(True, marks=pytest.mark.gpu)])
@pytest.mark.parametrize("proto", ["dmr", "ysf", "nxdn"])
def test_cli_pipe_like_the_example_scripts(oracle, tmp_path, proto, gpu):
    """examples/dmr-decoder.sh:19-23 / ysf-decoder.sh: rrc_filter | gfsk_demodulator | <proto>_decoder --fifo <meta>."""
    bindir = _build_tools(tmp_path, gpu)
    if proto == "nxdn":                                                   # examples/nxdn48-decoder.sh:19-23
        from digiham_amd import _taps
        s = synth.nxdn_stream(75, 16, src=4660, dst=99)
        x = synth.impair(synth.shape(s, sps=20, taps=_taps.narrow()), 7, snr_db=24, dc=0.02, delay=3)
    else:
        s = synth.dmr_stream(73, 30, two_slots=False) if proto == "dmr" else synth.ysf_stream(71, 8)
        x = synth.impair(synth.shape(s), 7, snr_db=24, dc=0.02, delay=3)
    inp, meta, out = tmp_path / "in.f32", tmp_path / "meta.txt", tmp_path / "out.bin"
    x.astype(np.float32).tofile(inp)
    opts = ("-n", "-s 20") if proto == "nxdn" else ("", "")
    cmd = "%s/rrc_filter %s < %s | %s/gfsk_demodulator %s | %s/%s_decoder --fifo %s > %s" % (bindir, opts[0], inp, bindir, opts[1], bindir, proto, meta, out)
    subprocess.run(["bash", "-o", "pipefail", "-c", cmd], check=True, stderr=subprocess.DEVNULL)
    # reference output for the same dibits: the tools consume their whole input, the oracle stops sps+1 samples early
    ref = oracle.chain(x[None, :], proto={"dmr": 1, "ysf": 2, "nxdn": 3}[proto], **(dict(rrc=2, sps=20) if proto == "nxdn" else {}))
    got = np.fromfile(out, np.uint8)
    want = ref["out"][0, :ref["out_count"][0]]
    assert len(want) > 0 and len(got) >= len(want) and (got[:len(want)] == want).all()
    lines = meta.read_text().splitlines()
    assert lines and all(l.startswith("protocol:%s" % proto.upper()) or ";protocol:%s" % proto.upper() in l for l in lines)
    if proto == "dmr":
        ev = ref["events"][0, :ref["event_count"][0]]
        lcs = ev[ev["type"] == 4]
        assert len(lcs) > 0
        lc = api.parse_lc(bytes(lcs[0]["payload"][:9]))
        assert any("source:%d;" % lc["source"] in l and "target:%d;type:group" % lc["target"] in l and "slot:%d" % lcs[0]["a"] in l
                   for l in lines), lines
    elif proto == "ysf":
        assert any("mode:DN;protocol:YSF" in l for l in lines), lines
    else:
        assert "destination:99;protocol:NXDN;source:4660;sync:voice;type:individual" in lines, lines


@pytest.mark.parametrize("gpu", [False, pytest.param(True, marks=pytest.mark.gpu)])
def test_pocsag_pipe_like_the_example_script(oracle, tmp_path, gpu):
    """examples/pocsag-decoder.sh: fsk_demodulator -i -s 40 | pocsag_decoder; the pages are the decoder's stdout."""
    bindir = _build_tools(tmp_path, gpu)
    bits, sent = synth.pocsag_stream(31, 3)
    x = synth.impair(synth.fsk_shape(bits, sps=40, invert=True), 5, snr_db=22, dc=0.03)
    inp, out = tmp_path / "in.f32", tmp_path / "out.txt"
    x.astype(np.float32).tofile(inp)
    cmd = "%s/fsk_demodulator -i -s 40 < %s | %s/pocsag_decoder > %s" % (bindir, inp, bindir, out)
    subprocess.run(["bash", "-o", "pipefail", "-c", cmd], check=True, stderr=subprocess.DEVNULL)
    ref = oracle.chain(x[None, :], rrc=0, levels=2, invert=True, sps=40, proto=4)
    want = ref["out"][0, :ref["out_count"][0]]
    got = np.fromfile(out, np.uint8)
    assert len(want) > 0 and len(got) >= len(want) and (got[:len(want)] == want).all()
    lines = bytes(got).decode("latin1").split("\n")
    assert any(("address:%d;message:%s" % (a, t)) in lines for a, f, t in sent)


@pytest.mark.parametrize("gpu", [False, pytest.param(True, marks=pytest.mark.gpu)])
def test_dstar_pipe_like_the_example_script(oracle, tmp_path, gpu):
    """examples/dstar-decoder.sh: fsk_demodulator -s 10 | dstar_decoder --fifo <meta>: voice frames on stdout, header fields,
    the slow-data message, a DPRS sentence and an NMEA position on the metadata pipe."""
    bindir = _build_tools(tmp_path, gpu)
    rng = np.random.default_rng(9)
    dprs = "DL1ABC>APDPRS,DSTAR*:!4916.45N/01131.00E>via d-star"
    t1, _, _ = synth.dstar_transmission(rng, my="DL1ABC", suffix="ID51", message="Jakob, JN68", n_superframes=5,
                                        simple=synth.dstar_dprs_sentence(dprs))
    t2, _, _ = synth.dstar_transmission(rng, my="W1AW", suffix="", your="DL1ABC", message="position follows", n_superframes=5,
                                        simple=synth.dstar_gga_sentence(-33.5, 151.25))
    bits = np.concatenate([rng.integers(0, 2, 100).astype(np.uint8), t1, rng.integers(0, 2, 300).astype(np.uint8), t2,
                           rng.integers(0, 2, 400).astype(np.uint8)])
    x = synth.impair(synth.fsk_shape(bits, sps=10), 5, snr_db=22, dc=0.03)
    inp, meta, out = tmp_path / "in.f32", tmp_path / "meta.txt", tmp_path / "out.bin"
    x.astype(np.float32).tofile(inp)
    cmd = "%s/fsk_demodulator -s 10 < %s | %s/dstar_decoder --fifo %s > %s" % (bindir, inp, bindir, meta, out)
    subprocess.run(["bash", "-o", "pipefail", "-c", cmd], check=True, stderr=subprocess.DEVNULL)
    ref = oracle.chain(x[None, :], rrc=0, levels=2, sps=10, proto=5)
    want = ref["out"][0, :ref["out_count"][0]]
    got = np.fromfile(out, np.uint8)
    assert len(want) > 200 * 9 and len(got) >= len(want) and (got[:len(want)] == want).all()
    lines = meta.read_text().splitlines()
    head = "departure:DB0XYZ B;destination:DB0XYZ G;ourcall:DL1ABC/ID51;protocol:DSTAR;sync:voice;yourcall:CQCQCQ"
    assert head in lines, lines
    assert "departure:DB0XYZ B;destination:DB0XYZ G;message:Jakob, JN68         ;ourcall:DL1ABC/ID51;protocol:DSTAR;sync:voice;yourcall:CQCQCQ" in lines
    assert any(";dprs:%s;" % dprs in l and "ourcall:DL1ABC/ID51" in l for l in lines), lines
    assert "protocol:DSTAR" in lines                                       # MetaCollector::reset() at the end pattern
    assert any("ourcall:W1AW;" in l and "yourcall:DL1ABC" in l for l in lines)
    pos = [l for l in lines if ";lat:" in l]
    assert pos and all("lat:-33.5" in l and "lon:151.25" in l for l in pos), pos


def test_cli_tools_fail_loudly_without_a_device():
    """Option parsing mirrors the reference's getopt tables; with no MI355X the tools exit non-zero with a message."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "cli"), "-s"], check=True)
    for t in TOOLS:
        exe = os.path.join(ROOT, "cli", "bin", t)
        v = subprocess.run([exe, "--version"], capture_output=True)
        assert v.returncode == 0 and v.stdout.decode().startswith(t + " version ")
        h = subprocess.run([exe, "-h"], capture_output=True).stderr.decode()
        assert "Usage: %s [options]" % t in h
    assert "--narrow" in subprocess.run([os.path.join(ROOT, "cli", "bin", "rrc_filter"), "-h"], capture_output=True).stderr.decode()
    assert "--control-fifo" in subprocess.run([os.path.join(ROOT, "cli", "bin", "dmr_decoder"), "-h"], capture_output=True).stderr.decode()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except ImportError:
        has_gpu = False
    if not has_gpu:
        r = subprocess.run([os.path.join(ROOT, "cli", "bin", "rrc_filter")], input=np.zeros(64, np.float32).tobytes(), capture_output=True)
        assert r.returncode != 0 and b"rrc_filter:" in r.stderr


# ---------------------------------------------------------------------------------------------------------------------
# Metadata lines from ENGINE events (CPU wave emulation / -m gpu: libdigiham_amd.so): a DMR call whose embedded signalling
# carries a talker alias and a GPS position, a YSF call whose data frames carry a GPS position.  What the lines must say
# comes from the REFERENCE's own classes (tests/golden/elements_ref.npz: TalkerAliasCollector, Dmr::Gps, Ysf::Gps).
def _dmr_call_with_embedded(rng, lcs, cc=1, dst=1234, src=5678901):
    """bursts of one voice call on slot 0 (idle bursts on slot 1): LC header, one superframe per entry of `lcs` whose
    embedded signalling carries that 9-byte LC, terminator"""
    head = synth.dmr_lc(0, 0, 0, dst, src)
    bursts = [synth.dmr_data_burst(0, cc, 1, head + bytes(3), "bs_data", rng)]
    for lc9 in lcs:
        frags = synth.dmr_embedded_lc_fragments(lc9)
        for f in range(6):
            mid = synth.DMR_SYNC["bs_voice"] if f == 0 else synth.dmr_emb_mid(cc, [1, 3, 3, 2][f - 1], frags[f - 1]) if f <= 4 \
                else synth.dmr_emb_mid(cc, 0, [0] * 16)
            bursts.append(synth.dmr_voice_burst(0, list(rng.integers(0, 4, 108)), mid, rng))
    bursts.append(synth.dmr_data_burst(0, cc, 2, head + bytes(3), "bs_data", rng))
    out = list(rng.integers(0, 4, 41))
    for b in bursts:
        out += b + synth.dmr_idle_burst(1, cc, rng)
    return np.array(out + [0] * 200, np.uint8)


def _lines_from_engine(ctx, tmp_path, proto, syms):
    eng = api.Engine(1, len(syms), rrc="none", demod="none", proto=proto, ctx=ctx)
    batches = []
    for lo in range(0, len(syms), 5000):                        # several decoder calls, as a pipe would deliver them
        part = np.ascontiguousarray(syms[None, lo:lo + 5000])
        eng.push_symbols(part, np.array([part.shape[1]], np.uint32))
        e, ec = eng.events()
        batches.append([e[0, i:i + 1] for i in range(ec[0])])
    eng.close()
    exe = _meta_test(tmp_path)
    return subprocess.run([exe, proto], input=_batches(batches), capture_output=True, check=True).stdout.split(b"\n")


def test_dmr_talker_alias_and_gps_lines_from_engine_events(ctx, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "elements_ref.npz"))
    # a golden talker alias vector: 8-bit format, all four blocks in order, complete, printable
    pick = [i for i in range(len(g["ta_len"])) if g["ta_blocks"][i, 0] >> 6 == 1 and tuple(g["ta_order"][i]) == (0, 1, 2, 3)
            and g["ta_complete"][i] & 8 and g["ta_len"][i] >= 20 and 0 not in g["ta_blocks"][i, 1:]][0]
    blocks = g["ta_blocks"][pick]
    alias = bytes(g["ta_text"][pick, :g["ta_len"][pick]])       # bytes: the reference cuts at a byte count, possibly inside a character
    k = 77
    gps_bytes, (lat, lon) = g["dmr_gps_in"][k], g["dmr_gps_out"][k]
    group = synth.dmr_lc(0, 0, 0, 1234, 5678901)
    lcs = [group] + [bytes([4 + b, 0]) + bytes(blocks[7 * b:7 * b + 7]) for b in range(4)] + [bytes([8, 0]) + bytes(gps_bytes), group]
    lines = _lines_from_engine(ctx, tmp_path, "dmr", _dmr_call_with_embedded(np.random.default_rng(3), lcs))
    slot0 = [l for l in lines if b";slot:0" in l]
    assert b"protocol:DMR;slot:0;source:5678901;sync:voice;target:1234;type:group" in slot0
    with_alias = b"protocol:DMR;slot:0;source:5678901;sync:voice;talkeralias:" + alias + b";target:1234;type:group"
    assert with_alias in slot0, slot0
    pos = ("lat:%f;lon:%f;" % (lat, lon)).encode()
    assert pos + with_alias in slot0, slot0
    assert slot0.index(with_alias) < slot0.index(pos + with_alias)
    assert slot0[-1] in (b"protocol:DMR;slot:0;sync:data", b"protocol:DMR;slot:0")      # the terminator ends the call


def test_ysf_gps_lines_from_engine_events(ctx, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "elements_ref.npz"))
    k = int(np.nonzero(g["ysf_gps_ok"])[0][5])
    lat, lon = g["ysf_gps_out"][k]
    dt = bytearray(20)
    dt[1:4] = bytes([0x22, 0x62, 0x5F])                                   # COMMAND_SHORT_GPS (data.hpp:8)
    dt[4] = 0x28
    dt[5:14] = bytes(g["ysf_gps_in"][k])
    dt[18] = 0x03
    dt[19] = sum(dt[:19]) & 0xFF
    rng = np.random.default_rng(4)
    dch = {0: b"CQCQCQ    ", 1: b"DL1ABC    ", 2: b"DB0XYZ    ", 3: b"DB0XYZ    ", 6: bytes(dt[:10]), 7: bytes(dt[10:])}
    out = list(rng.integers(0, 4, 53)) + synth.ysf_frame(rng, 0, 2)
    for fn in range(8):
        out += synth.ysf_frame(rng, 1, 2, fn, dch=dch.get(fn))
    out += synth.ysf_frame(rng, 2, 2) + [0] * 600
    lines = _lines_from_engine(ctx, tmp_path, "ysf", np.array(out, np.uint8))
    assert any((";lat:%f;lon:%f;mode:DN;protocol:YSF" % (lat, lon)).encode() in b";" + l for l in lines), lines
    assert any(b"source:DL1ABC" in l and b"target:CQCQCQ" in l for l in lines)


def test_dmr_decoder_control_fifo_takes_commands_and_lets_the_tool_exit(oracle, tmp_path):
    """dmr_decoder -c <fifo> (src/dmr_decoder/dmr_cli.cpp:57-78): a slot filter written to the fifo arrives, a fifo nobody writes
    to -- or whose writer sits idle -- does not keep the tool from exiting at the end of its input (the reference hangs there),
    and a second -c is ignored."""
    import time
    bindir = tmp_path / "bin"
    bindir.mkdir()
    import hostemu
    hostemu.build()
    libdir = os.path.join(ROOT, "tests", "host_harness")
    exe = str(bindir / "dmr_decoder")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "cli", "dmr_decoder.cpp"),
                    "-o", exe, "-L" + libdir, "-ldh_hostemu", "-Wl,-rpath," + libdir, "-pthread"], check=True)
    fifo = str(tmp_path / "ctl.fifo")
    os.mkfifo(fifo)
    syms = synth.dmr_stream(91, 30, two_slots=True)
    # (1) no writer at all: the tool must come back at the end of its input
    t0 = time.time()
    r = subprocess.run([exe, "-c", fifo, "-c", fifo], input=syms.tobytes(), capture_output=True, timeout=20)
    assert r.returncode == 0 and time.time() - t0 < 10
    both = np.frombuffer(r.stdout, np.uint8)
    # (2) a writer that sets the filter first and then sits idle with the fifo open
    p = subprocess.Popen([exe, "-c", fifo], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    w = os.open(fifo, os.O_WRONLY)
    os.write(w, b"2\n")
    time.sleep(0.5)                                               # the control thread has seen it before the symbols arrive
    out, _ = p.communicate(syms.tobytes(), timeout=20)
    os.close(w)
    assert p.returncode == 0
    d = oracle.Decoder("dmr"); d.set_slot_filter(2)
    want, _ = d.process(syms)
    got = np.frombuffer(out, np.uint8)
    assert len(want) > 0 and len(got) == len(want) and (got == want).all()
    assert got.tobytes() != both.tobytes()                        # (unfiltered, the first slot to speak wins)
