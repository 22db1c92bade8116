#!/usr/bin/env python3
"""Known-answer vectors for the BLS12-381 scalar-field transforms -> tests/golden/bls12_381_ntt_vectors.json.

The reference has no such field (curve.py:2: BN254 throughout), so these vectors are NOT reference outputs: they are the
definition evaluated in plain Python integers by code that shares nothing with the C oracle or the kernels —
  * the O(n^2) sum X[k] = sum_j x[j] w^(jk) at n = 2^3 and 2^6, w = 7^((r-1)/n);
  * above that a recursive even/odd split checked against that sum at 2^6, for n = 2^8 .. 2^13 and 2^16;
  * the coset forms of poly.py:156-177 over this field spelled out with it.
Inputs are seeded (random.Random(seed).randrange(r)); outputs are stored as head / tail / SHA-256 summaries (tests/helpers.py).
Pins oracle/c's oracle_bls_fr_ntt (tests/test_oracle_c.py) and, through the same file, plonk_bls_fr_* on the GPU."""
import hashlib
import json
import os
import random
import sys

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ROOT_2_32 = 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B  # the bls12_381 crate's ROOT_OF_UNITY
assert pow(7, (R - 1) >> 32, R) == ROOT_2_32 and pow(ROOT_2_32, 1 << 31, R) == R - 1


def dft_sum(xs, w):
    n = len(xs)
    return [sum(x * pow(w, j * k, R) for j, x in enumerate(xs)) % R for k in range(n)]


def fft(xs, w):
    n = len(xs)
    if n == 1:
        return list(xs)
    ev, od = fft(xs[0::2], w * w % R), fft(xs[1::2], w * w % R)
    out, t = [0] * n, 1
    for k in range(n // 2):
        u = od[k] * t % R
        out[k], out[k + n // 2] = (ev[k] + u) % R, (ev[k] - u) % R
        t = t * w % R
    return out


def ntt(xs, inverse=False):
    n = len(xs)
    w = pow(7, (R - 1) // n, R)
    if not inverse:
        return fft(xs, w)
    ninv = pow(n, -1, R)
    return [v * ninv % R for v in fft(xs, pow(w, -1, R))]


def summary(ints):
    h = hashlib.sha256()
    for v in ints:
        h.update(int(v).to_bytes(32, "big"))
    return {"n": len(ints), "head": [str(v) for v in ints[:4]], "tail": [str(v) for v in ints[-4:]], "sha256_be32": h.hexdigest()}


def main(out):
    for n in (8, 64):
        xs = [random.Random(n).randrange(R) for _ in range(n)]
        assert fft(xs, pow(7, (R - 1) // n, R)) == dft_sum(xs, pow(7, (R - 1) // n, R))
    cases = []
    for log_n in (8, 9, 10, 11, 12, 13, 16):
        seed = 381000 + log_n
        rng = random.Random(seed)
        xs = [rng.randrange(R) for _ in range(1 << log_n)]
        case = {"log_n": log_n, "seed": seed, "fft": summary(ntt(xs)), "ifft": summary(ntt(xs, True))}
        if log_n <= 11:  # poly.py:156-163 / 169-177 over this field
            off = rng.randrange(2, R)
            coeffs = ntt(xs, True)
            scaled = [c * pow(off, i, R) % R for i, c in enumerate(coeffs)] + [0] * (3 << log_n)
            case["offset"] = str(off)
            case["coset_extend"] = summary(ntt(scaled))
            back = ntt(xs, True)  # coset_to_coeffs of 2^log_n coset values: ifft, then v_i / off^i
            oinv = pow(off, -1, R)
            case["coset_to_coeffs"] = summary([v * pow(oinv, i, R) % R for i, v in enumerate(back)])
        cases.append(case)
    json.dump({"source": "tools/gen_bls_vectors.py: the DFT definition over the BLS12-381 scalar field in Python integers (no reference "
                         "counterpart; generator 7, root of unity = the bls12_381 crate's ROOT_OF_UNITY squared down)",
               "modulus": hex(R), "cases": cases}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bls12_381_ntt_vectors.json"))
