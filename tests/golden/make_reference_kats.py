"""Extract the known-answer DATA (numbers only) held by the reference's own unit tests into tests/golden/*.json.

Run in the build container (needs /root/reference; it is never read at test time):
    python tests/golden/make_reference_kats.py
Only literal numbers are taken -- coordinate lists and expected values -- no reference source text is kept.
Sources (relative to /root/reference):
  mundy/math/tests/unit_tests/UnitTestHilbert.cpp:48-387   expected Hilbert positions (s=2,4,8) and directors (8/9 links)
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
VEC = re.compile(r"Vector3d\(\s*(-?[0-9.eE+-]+)\s*,\s*(-?[0-9.eE+-]+)\s*,\s*(-?[0-9.eE+-]+)\s*\)")


def vec_list(block, name):
    """all Vector3d(...) triples of the initializer list of `name` inside `block`"""
    start = block.index(name)
    end = block.index("};", start)
    return [[float(a), float(b), float(c)] for a, b, c in VEC.findall(block[start:end])]


def hilbert():
    src = open(os.path.join(REF, "mundy/math/tests/unit_tests/UnitTestHilbert.cpp")).read()
    tests = re.split(r"\nTEST\(Hilbert3D, ", src)[1:]
    out = {}
    for t in tests:
        name = t[: t.index(")")]
        entry = {"positions": vec_list(t, "expected_position_array")}
        if "expected_directors" in t:
            entry["directors"] = vec_list(t, "expected_directors")
        out[name] = entry
    return out


if __name__ == "__main__":
    h = hilbert()
    assert [len(h[k]["positions"]) for k in ("Cube2", "Cube4", "Cube8")] == [8, 64, 512], {k: len(v["positions"]) for k, v in h.items()}
    with open(os.path.join(HERE, "hilbert_kat.json"), "w") as f:
        json.dump(h, f, separators=(",", ":"))
    print({k: {kk: len(vv) for kk, vv in v.items()} for k, v in h.items()})
