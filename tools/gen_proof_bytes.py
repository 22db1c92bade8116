#!/usr/bin/env python3
"""tests/golden/k6_proof_bytes.json: the 480-byte compressed form of the reference's golden proof.

The VALUES are the reference's own (test/proof.pickle as committed in tests/golden/k6_proof.json); the reference has no
compressed encoding (its only G1 bytes are x, y as 32-byte big-endian integers, transcript.py:62-67), so the BYTES follow
this build's definition (plonkathon_amd/csrc/g1_codec.h), spelled out here independently of the product and of oracle/:
  point  = x as 32 big-endian bytes, byte 0 |= 0x80 if y <= (p-1)/2 else 0xC0   (0x40 and x = 0 for the identity)
  scalar = 32 big-endian bytes
  proof  = a_1 b_1 c_1 z_1 t_lo_1 t_mid_1 t_hi_1 W_z_1 W_zw_1 | a_eval b_eval c_eval s1_eval s2_eval z_shifted_eval"""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583

proof = json.load(open(os.path.join(GOLDEN, "k6_proof.json")))["proof"]
out = b""
for k in ("a_1", "b_1", "c_1", "z_1", "t_lo_1", "t_mid_1", "t_hi_1", "W_z_1", "W_zw_1"):
    v = proof[k]
    if v is None:
        out += bytes([0x40]) + bytes(31)
        continue
    x, y = int(v[0]), int(v[1])
    b = bytearray(x.to_bytes(32, "big"))
    b[0] |= 0xC0 if y > (P - 1) // 2 else 0x80
    out += bytes(b)
for k in ("a_eval", "b_eval", "c_eval", "s1_eval", "s2_eval", "z_shifted_eval"):
    out += int(proof[k]).to_bytes(32, "big")
assert len(out) == 480
json.dump({"source": "values: test/proof.pickle (tests/golden/k6_proof.json); encoding: this build's (csrc/g1_codec.h), "
                     "written out by tools/gen_proof_bytes.py",
           "hex": out.hex(), "sha256": hashlib.sha256(out).hexdigest()},
          open(os.path.join(GOLDEN, "k6_proof_bytes.json"), "w"), indent=1)
print(hashlib.sha256(out).hexdigest())
