/* bn254_oracle.c — plain-C restatement of the two heavy pieces of the path, for parity checks at
 * sizes the pure-Python oracle does not finish in seconds.  TEST INFRASTRUCTURE ONLY (see
 * oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 *   oracle_fr_ntt   Polynomial.fft / ifft, /root/reference/poly.py:113-148: plain DFT
 *                   X[k] = sum_j x[j] w^(jk), w = 5^((r-1)/N) (curve.py:14-16), natural order in and
 *                   out; the inverse uses w^-1 and multiplies by 1/N (poly.py:131-139).
 *   oracle_bls_fr_ntt  the same transform over the BLS12-381 scalar field (generator 7); see its definition below.
 *   oracle_g1_lincomb  ec_lincomb, /root/reference/curve.py:38-44 (the `Equivalent to:` form at
 *                   curve.py:45-49: o = add(o, multiply(pt, coeff))), on the py_ecc group law.
 *
 * Deliberately a different algorithm from both the GPU kernels (no Montgomery windows/buckets/LDS
 * tiling) and the Python oracle (iterative in-place radix-2, Jacobian double-and-add), validated
 * against the Python oracle — and through it the reference's golden vectors — in
 * tests/test_oracle_c.py.  Elements are canonical, 4 x u64 little-endian.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;

static const fe FR_MOD = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static const fe FQ_MOD = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};

static int fe_geq(const fe* a, const fe* b) {
    for (int i = 3; i >= 0; i--) {
        if (a->v[i] > b->v[i]) return 1;
        if (a->v[i] < b->v[i]) return 0;
    }
    return 1;
}
static void fe_sub_raw(fe* r, const fe* a, const fe* b) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 x = (u128)a->v[i] - b->v[i] - br;
        r->v[i] = (uint64_t)x;
        br = (x >> 64) & 1;
    }
}
static void fe_add(fe* r, const fe* a, const fe* b, const fe* m) {
    u128 c = 0;
    fe t;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->v[i] + b->v[i];
        t.v[i] = (uint64_t)c;
        c >>= 64;
    }
    if (fe_geq(&t, m)) fe_sub_raw(&t, &t, m);
    *r = t;
}
static void fe_sub(fe* r, const fe* a, const fe* b, const fe* m) {
    fe t;
    if (fe_geq(a, b)) {
        fe_sub_raw(&t, a, b);
    } else {
        fe u;
        fe_sub_raw(&u, m, b);
        fe_add(&t, a, &u, m);
    }
    *r = t;
}
/* schoolbook 256x256 -> 512, then reduction by shift-subtract on 64-bit words (Knuth-free: we use
 * the simple fact 2^256 mod m is known through repeated doubling — slow but obviously correct). */
static void fe_mul(fe* r, const fe* a, const fe* b, const fe* m) {
    uint64_t p[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->v[i] * b->v[j] + p[i + j];
            p[i + j] = (uint64_t)c;
            c >>= 64;
        }
        p[i + 4] = (uint64_t)c;
    }
    /* Horner over the 8 words from the top: acc = acc * 2^64 + word (mod m), 64 doublings each */
    fe acc = {{0, 0, 0, 0}};
    for (int w = 7; w >= 0; w--) {
        for (int bit = 63; bit >= 0; bit--) {
            fe_add(&acc, &acc, &acc, m);
            if ((p[w] >> bit) & 1) {
                fe one = {{1, 0, 0, 0}};
                fe_add(&acc, &acc, &one, m);
            }
        }
    }
    *r = acc;
}
/* The bit-serial reduction above costs ~512 additions per product; fine for MSM-sized checks but too slow
 * for 2^20-point NTTs, so the NTT uses Montgomery arithmetic derived at run time from fe_mul. */
typedef struct { fe m; uint64_t ninv; fe r2; fe one; } mont;
static void mont_init(mont* M, const fe* m) {
    M->m = *m;
    uint64_t x = 1;  /* Newton: x = m^-1 mod 2^64 */
    for (int i = 0; i < 6; i++) x *= 2 - m->v[0] * x;
    M->ninv = (uint64_t)0 - x;
    fe t = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; i++) fe_add(&t, &t, &t, m); /* 2^256 mod m */
    M->one = t;
    fe_mul(&M->r2, &t, &t, m);
}
static void mont_mul(fe* r, const fe* a, const fe* b, const mont* M) {
    uint64_t t[6] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->v[j] * b->v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * M->ninv;
        c = ((u128)q * M->m.v[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)q * M->m.v[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&o, &M->m)) fe_sub_raw(&o, &o, &M->m);
    *r = o;
}
static void mont_pow(fe* r, const fe* a, const fe* e, const mont* M) {
    fe acc = M->one;
    for (int w = 3; w >= 0; w--)
        for (int bit = 63; bit >= 0; bit--) {
            mont_mul(&acc, &acc, &acc, M);
            if ((e->v[w] >> bit) & 1) mont_mul(&acc, &acc, a, M);
        }
    *r = acc;
}

/* ---- exported: field helpers for the tests ---------------------------------------------------- */
void oracle_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    fe r;
    fe_mul(&r, (const fe*)a, (const fe*)b, &FR_MOD);
    memcpy(out, r.v, 32);
}

/* ---- exported: NTT ---------------------------------------------------------------------------- */
/* The transform over any prime field of up to 256 bits: w = generator^((m-1)/n). */
static int ntt_over(uint64_t* data, unsigned log_n, int inverse, const fe* modulus, uint64_t generator, unsigned two_adicity) {
    if (log_n > two_adicity) return -1;
    const size_t n = (size_t)1 << log_n;
    mont M;
    mont_init(&M, modulus);
    fe* x = (fe*)data;
    for (size_t i = 0; i < n; i++) mont_mul(&x[i], &x[i], &M.r2, &M); /* to Montgomery */
    /* w = generator^((m-1)/n), or its inverse */
    fe five = {{generator, 0, 0, 0}}, e, w;
    mont_mul(&five, &five, &M.r2, &M);
    fe rm1;
    fe one_raw = {{1, 0, 0, 0}};
    fe_sub_raw(&rm1, modulus, &one_raw);
    /* e = (r-1) >> log_n */
    e = rm1;
    for (unsigned s = 0; s < log_n; s++) {
        for (int i = 0; i < 4; i++) e.v[i] = (e.v[i] >> 1) | (i < 3 ? e.v[i + 1] << 63 : 0);
    }
    mont_pow(&w, &five, &e, &M);
    if (inverse) {
        fe rm2, two_raw = {{2, 0, 0, 0}};
        fe_sub_raw(&rm2, modulus, &two_raw);
        mont_pow(&w, &w, &rm2, &M);
    }
    /* bit reversal */
    for (size_t i = 0, j = 0; i < n; i++) {
        if (i < j) { fe t = x[i]; x[i] = x[j]; x[j] = t; }
        size_t bit = n >> 1;
        for (; bit && (j & bit); bit >>= 1) j ^= bit;
        j |= bit;
    }
    /* twiddle table w^0 .. w^(n/2-1) */
    fe* tw = (fe*)malloc((n / 2 ? n / 2 : 1) * sizeof(fe));
    if (!tw) return -2;
    tw[0] = M.one;
    for (size_t i = 1; i < n / 2; i++) mont_mul(&tw[i], &tw[i - 1], &w, &M);
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1, step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t j = 0; j < half; j++) {
                fe t, u = x[i + j];
                mont_mul(&t, &x[i + j + half], &tw[j * step], &M);
                fe_add(&x[i + j], &u, &t, modulus);
                fe_sub(&x[i + j + half], &u, &t, modulus);
            }
    }
    free(tw);
    fe scale = {{1, 0, 0, 0}}; /* from Montgomery: multiply by 1; inverse also folds in 1/n */
    if (inverse) {
        fe nn = {{(uint64_t)n, 0, 0, 0}}, rm2, two_raw = {{2, 0, 0, 0}};
        mont_mul(&nn, &nn, &M.r2, &M);
        fe_sub_raw(&rm2, modulus, &two_raw);
        mont_pow(&nn, &nn, &rm2, &M);          /* 1/n in Montgomery form */
        mont_mul(&scale, &nn, &scale, &M);     /* -> canonical 1/n */
        for (size_t i = 0; i < n; i++) {
            fe t;
            mont_mul(&t, &x[i], &scale, &M);   /* x_mont * (1/n)_canonical * R^-1 = canonical x/n ... see below */
            x[i] = t;
        }
        /* x[i] was a*R; a*R * (1/n) * R^-1 = a/n canonical. */
    } else {
        for (size_t i = 0; i < n; i++) mont_mul(&x[i], &x[i], &scale, &M);
    }
    return 0;
}

int oracle_fr_ntt(uint64_t* data, unsigned log_n, int inverse) { return ntt_over(data, log_n, inverse, &FR_MOD, 5, 28); }

/* The same transform over the BLS12-381 scalar field (the field of BASELINE.json's standalone-NTT metric; the reference
 * itself has no such field — curve.py:2 is BN254 throughout): generator 7, 2-adicity 32, so that w_{2^32} = 7^((r-1)/2^32)
 * is the ROOT_OF_UNITY constant of the `bls12_381` crate (checked in tests/test_oracle_c.py).  Pinned by definition only:
 * against the O(n^2) DFT sum in Python integers at small sizes. */
static const fe BLS_FR_MOD = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
int oracle_bls_fr_ntt(uint64_t* data, unsigned log_n, int inverse) { return ntt_over(data, log_n, inverse, &BLS_FR_MOD, 7, 32); }

/* ---- exported: G1 linear combination (Jacobian double-and-add, a = 0, b = 3) ------------------- */
typedef struct { fe x, y, z; } jac; /* z == 0 <=> identity */
static mont MQ;
static int mq_ready = 0;
static void jac_dbl(jac* r, const jac* p) {
    if ((p->z.v[0] | p->z.v[1] | p->z.v[2] | p->z.v[3]) == 0) { *r = *p; return; }
    fe a, b, c, d, e, f, t;
    mont_mul(&a, &p->x, &p->x, &MQ);
    mont_mul(&b, &p->y, &p->y, &MQ);
    mont_mul(&c, &b, &b, &MQ);
    fe_add(&t, &p->x, &b, &FQ_MOD); mont_mul(&t, &t, &t, &MQ);
    fe_sub(&t, &t, &a, &FQ_MOD); fe_sub(&t, &t, &c, &FQ_MOD); fe_add(&d, &t, &t, &FQ_MOD);
    fe_add(&e, &a, &a, &FQ_MOD); fe_add(&e, &e, &a, &FQ_MOD);
    mont_mul(&f, &e, &e, &MQ);
    jac o;
    fe_sub(&o.x, &f, &d, &FQ_MOD); fe_sub(&o.x, &o.x, &d, &FQ_MOD);
    fe c8; fe_add(&c8, &c, &c, &FQ_MOD); fe_add(&c8, &c8, &c8, &FQ_MOD); fe_add(&c8, &c8, &c8, &FQ_MOD);
    fe_sub(&t, &d, &o.x, &FQ_MOD); mont_mul(&t, &e, &t, &MQ); fe_sub(&o.y, &t, &c8, &FQ_MOD);
    mont_mul(&t, &p->y, &p->z, &MQ); fe_add(&o.z, &t, &t, &FQ_MOD);
    *r = o;
}
static void jac_add(jac* r, const jac* p, const jac* q) {
    if ((p->z.v[0] | p->z.v[1] | p->z.v[2] | p->z.v[3]) == 0) { *r = *q; return; }
    if ((q->z.v[0] | q->z.v[1] | q->z.v[2] | q->z.v[3]) == 0) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
    mont_mul(&z1z1, &p->z, &p->z, &MQ);
    mont_mul(&z2z2, &q->z, &q->z, &MQ);
    mont_mul(&u1, &p->x, &z2z2, &MQ);
    mont_mul(&u2, &q->x, &z1z1, &MQ);
    mont_mul(&t, &q->z, &z2z2, &MQ); mont_mul(&s1, &p->y, &t, &MQ);
    mont_mul(&t, &p->z, &z1z1, &MQ); mont_mul(&s2, &q->y, &t, &MQ);
    fe_sub(&h, &u2, &u1, &FQ_MOD);
    fe_sub(&rr, &s2, &s1, &FQ_MOD);
    if ((h.v[0] | h.v[1] | h.v[2] | h.v[3]) == 0) {
        if ((rr.v[0] | rr.v[1] | rr.v[2] | rr.v[3]) == 0) { jac_dbl(r, p); return; }
        memset(r, 0, sizeof *r);
        return;
    }
    fe hh, hhh, v;
    mont_mul(&hh, &h, &h, &MQ);
    mont_mul(&hhh, &h, &hh, &MQ);
    mont_mul(&v, &u1, &hh, &MQ);
    jac o;
    mont_mul(&t, &rr, &rr, &MQ);
    fe_sub(&t, &t, &hhh, &FQ_MOD); fe_sub(&t, &t, &v, &FQ_MOD); fe_sub(&o.x, &t, &v, &FQ_MOD);
    fe_sub(&t, &v, &o.x, &FQ_MOD); mont_mul(&t, &rr, &t, &MQ);
    fe t2; mont_mul(&t2, &s1, &hhh, &MQ); fe_sub(&o.y, &t, &t2, &FQ_MOD);
    mont_mul(&t, &p->z, &q->z, &MQ); mont_mul(&o.z, &t, &h, &MQ);
    *r = o;
}

/* points: n x (x, y) canonical, (0,0) = identity; scalars: n canonical values (already < r) */
int oracle_g1_lincomb(const uint64_t* points_xy, const uint64_t* scalars, size_t n, uint64_t out_xy[8], int* is_identity) {
    if (!mq_ready) { mont_init(&MQ, &FQ_MOD); mq_ready = 1; }
    jac acc;
    memset(&acc, 0, sizeof acc);
    for (size_t i = 0; i < n; i++) {
        const fe* px = (const fe*)(points_xy + 8 * i);
        const fe* py = px + 1;
        const fe* k = (const fe*)(scalars + 4 * i);
        if ((px->v[0] | px->v[1] | px->v[2] | px->v[3] | py->v[0] | py->v[1] | py->v[2] | py->v[3]) == 0) continue;
        jac base, term;
        mont_mul(&base.x, px, &MQ.r2, &MQ);
        mont_mul(&base.y, py, &MQ.r2, &MQ);
        base.z = MQ.one;
        memset(&term, 0, sizeof term);
        for (int w = 3; w >= 0; w--)
            for (int bit = 63; bit >= 0; bit--) {
                jac_dbl(&term, &term);
                if ((k->v[w] >> bit) & 1) jac_add(&term, &term, &base);
            }
        jac_add(&acc, &acc, &term);
    }
    if ((acc.z.v[0] | acc.z.v[1] | acc.z.v[2] | acc.z.v[3]) == 0) {
        *is_identity = 1;
        memset(out_xy, 0, 64);
        return 0;
    }
    *is_identity = 0;
    fe zi, zi2, zi3, qm2, two_raw = {{2, 0, 0, 0}}, one_raw = {{1, 0, 0, 0}}, x, y;
    fe_sub_raw(&qm2, &FQ_MOD, &two_raw);
    mont_pow(&zi, &acc.z, &qm2, &MQ);
    mont_mul(&zi2, &zi, &zi, &MQ);
    mont_mul(&zi3, &zi2, &zi, &MQ);
    mont_mul(&x, &acc.x, &zi2, &MQ);
    mont_mul(&y, &acc.y, &zi3, &MQ);
    mont_mul(&x, &x, &one_raw, &MQ);
    mont_mul(&y, &y, &one_raw, &MQ);
    memcpy(out_xy, x.v, 32);
    memcpy(out_xy + 4, y.v, 32);
    return 0;
}
