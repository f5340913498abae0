#!/usr/bin/env python3
"""Digest of the reference's marching-cubes tables, taken from its source files WHERE THEY LIE (this container only; the
reference tree is not on the GPU box and none of its text is copied):

    python tools/mc_table_sha.py            # prints the digests and compares them with tests/golden/mc_tables.sha256.json
    python tools/mc_table_sha.py --write    # (re)writes that file

Digests: SHA-256 over the row-major bytes of TRIANGLE_TABLE[256][16] as int8 (MC_triangle_table.cu:87),
VERTICES_FOR_CUBE_TYPE[256] as uint8 (:46) and EDGE_VERTICES[12][2] as uint8 (MC_edge_table.cu:47).  Only the digests are
committed; tests/test_host_marching_cubes.py and tests/test_oracle_pins.py hash the product's and the oracle's tables
(built from base configurations + rotations) and compare."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("REF", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden", "mc_tables.sha256.json")


def numbers_after(text, marker):
    body = text[text.index(marker):]
    body = body[body.index("=") + 1:body.index(";")]
    return [int(x, 0) for x in re.findall(r"-?(?:0x[0-9a-fA-F]+|\d+)", body)]


def digests():
    tri_src = open(os.path.join(REF, "src/MarchingCubes/MC_triangle_table.cu")).read()
    edge_src = open(os.path.join(REF, "src/MarchingCubes/MC_edge_table.cu")).read()
    tri = numbers_after(tri_src, "TRIANGLE_TABLE[256][16]")
    cnt = numbers_after(tri_src, "VERTICES_FOR_CUBE_TYPE[256]")
    ev = numbers_after(edge_src, "EDGE_VERTICES[12][2]")
    assert len(tri) == 256 * 16 and len(cnt) == 256 and len(ev) == 24, (len(tri), len(cnt), len(ev))
    h = lambda vals: hashlib.sha256(bytes(v & 0xFF for v in vals)).hexdigest()
    return {"TRIANGLE_TABLE[256][16] int8": h(tri), "VERTICES_FOR_CUBE_TYPE[256] uint8": h(cnt), "EDGE_VERTICES[12][2] uint8": h(ev)}


if __name__ == "__main__":
    d = digests()
    print(json.dumps(d, indent=1))
    if "--write" in sys.argv:
        json.dump(d, open(GOLD, "w"), indent=1)
        print("wrote", GOLD)
    elif os.path.exists(GOLD):
        same = json.load(open(GOLD)) == d
        print("golden file", "matches" if same else "DIFFERS")
        sys.exit(0 if same else 1)
