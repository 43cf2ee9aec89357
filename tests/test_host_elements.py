"""The host-side element parsers of include/digiham/ (C++) against the REFERENCE's own classes: Dmr::Gps,
Dmr::TalkerAliasCollector, Dmr::Lc, Ysf::Gps (tests/golden/elements_ref.npz, produced by src/dmr_decoder/{gps,talkeralias,
lc}.cpp, src/ysf_decoder/gps.cpp, src/lib/{coordinate,charset}.cpp compiled in place -- oracle/Makefile `ref`), and the
ISO-8859-1 converter against ICU's result as recorded in the D-Star header strings."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("host") / "elements_test")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host_cpp", "elements_test.cpp"), "-o", out], check=True)
    return out


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "elements_ref.npz"))


def _run(exe, what, data):
    return subprocess.run([exe, what], input=np.ascontiguousarray(data, np.uint8).tobytes(), capture_output=True, check=True).stdout


def test_dmr_gps_vs_reference(exe, gold):
    out = np.frombuffer(_run(exe, "dmr_gps", gold["dmr_gps_in"]), np.float32).reshape(-1, 2)
    assert out.view(np.uint32).tobytes() == gold["dmr_gps_out"].view(np.uint32).tobytes()


def test_talker_alias_vs_reference(exe, gold):
    """all four formats x every announced length x complete / partial / out-of-order block sequences (3 072 vectors)"""
    inp = np.concatenate([gold["ta_blocks"], gold["ta_order"]], axis=1)
    out = np.frombuffer(_run(exe, "talkeralias", inp), np.uint8).reshape(-1, 66)
    assert (out[:, 0] == gold["ta_complete"]).all()
    assert (out[:, 1] == gold["ta_len"]).all()
    assert (out[:, 2:] == gold["ta_text"]).all()
    assert (gold["ta_len"] > 0).sum() > 2000


def test_lc_getters_vs_reference(exe, gold):
    raw = _run(exe, "lc", gold["lc_in"])
    rec = np.frombuffer(raw, np.dtype([("f", "<u4", (4,)), ("d", "u1", (7,))]))
    assert (rec["f"] == gold["lc_fields"]).all() and (rec["d"] == gold["lc_data7"]).all()


def test_ysf_gps_vs_reference(exe, gold):
    rec = np.frombuffer(_run(exe, "ysf_gps", gold["ysf_gps_in"]), np.dtype([("ok", "u1"), ("ll", "<f4", (2,))]))
    assert (rec["ok"] == gold["ysf_gps_ok"]).all()
    assert rec["ll"].view(np.uint32).tobytes() == gold["ysf_gps_out"].view(np.uint32).tobytes()
    assert 1000 < int(gold["ysf_gps_ok"].sum()) < len(rec)


def test_latin1_converter_vs_icu(exe, gold):
    """Converter::convertToUtf8 (charset.cpp:10-27, through ICU in the reference): the call-sign fields of the reference's
    D-Star Header::toString() (header.cpp:159-188) are the ICU conversions of header bytes 3..38, right-trimmed."""
    ok = gold["dh_ok"] == 1
    data, text = gold["dh_data"][ok], gold["dh_text"][ok]
    fields = []
    for h in data:
        for at, ln in ((3, 8), (11, 8), (19, 8), (27, 8), (35, 4)):
            f = np.zeros(16, np.uint8); f[:ln] = h[at:at + ln]
            fields.append(f)
    out = np.frombuffer(_run(exe, "latin1", np.array(fields)), np.uint8).reshape(-1, 33)
    conv = [bytes(r[1:1 + r[0]]).decode("utf-8").rstrip(" ") for r in out]
    n_special = 0
    for i, t in enumerate(text):
        dst, dpt, comp, own, suffix = conv[5 * i:5 * i + 5]
        expect = 'DST RPT: "%s" DPT RPT: "%s" COMPANION: "%s" CALLSIGN: "%s" ' % (dst, dpt, comp, own + ("/" + suffix if suffix else ""))
        assert bytes(t).rstrip(b"\0").decode("utf-8") == expect
        n_special += any(ord(c) > 127 for c in expect)
    assert n_special > 20
