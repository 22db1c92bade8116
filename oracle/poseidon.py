"""Mini-Poseidon hash and the PLONK program that proves one execution of it.
(oracle: test infrastructure only)

Follows /root/reference/test/mini_poseidon.py:19-42 (hash; round constants are the data file
test/poseidon_rc.json, kept as tests/golden/poseidon_rc.json) and the program generator
`output_proof_lang` at /root/reference/test.py:216-239.  Pinned by K7.
"""
import json
import os

from .field import R_MOD, inv

_RC_PATH = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "poseidon_rc.json")


def round_constants():
    with open(_RC_PATH) as f:
        return [[int(a) % R_MOD, int(b) % R_MOD, int(c) % R_MOD] for (a, b, c) in json.load(f)]


MDS = [inv(i) for i in range(3, 8)]  # mini_poseidon.py:24


def poseidon_hash(in1, in2, rc=None):  # mini_poseidon.py:27-42
    rc = rc or round_constants()
    L, M, R = in1 % R_MOD, in2 % R_MOD, 0
    for i in range(64):
        L = pow(L + rc[i][0], 5, R_MOD)
        M = (M + rc[i][1]) % R_MOD
        R = (R + rc[i][2]) % R_MOD
        if i < 4 or i >= 60:
            M = pow(M, 5, R_MOD)
            R = pow(R, 5, R_MOD)
        L, M, R = (
            (L * MDS[0] + M * MDS[1] + R * MDS[2]) % R_MOD,
            (L * MDS[1] + M * MDS[2] + R * MDS[3]) % R_MOD,
            (L * MDS[2] + M * MDS[3] + R * MDS[4]) % R_MOD,
        )
    return M


def poseidon_program_lines(rc=None):  # test.py:216-239
    rc = rc or round_constants()
    o = ["L0 public", "M0 public", "M64 public", "R0 <== 0"]
    for i in range(64):
        for j, pos in enumerate(("L", "M", "R")):
            if i < 4 or i >= 60 or pos == "L":
                o.append("%sadj%d <== %s%d + %d" % (pos, i, pos, i, rc[i][j]))
                o.append("%ssq%d <== %sadj%d * %sadj%d" % (pos, i, pos, i, pos, i))
                o.append("%sqd%d <== %ssq%d * %ssq%d" % (pos, i, pos, i, pos, i))
                o.append("%sqn%d <== %sqd%d * %sadj%d" % (pos, i, pos, i, pos, i))
            else:
                o.append("%sqn%d <== %s%d + %d" % (pos, i, pos, i, rc[i][j]))
        for j, pos in enumerate(("L", "M", "R")):
            o.append("%ssuma%d <== Lqn%d * %d" % (pos, i, i, MDS[j]))
            o.append("%ssumb%d <== %ssuma%d + Mqn%d * %d" % (pos, i, pos, i, i, MDS[j + 1]))
            o.append("%s%d <== %ssumb%d + Rqn%d * %d" % (pos, i + 1, pos, i, i, MDS[j + 2]))
    return o
