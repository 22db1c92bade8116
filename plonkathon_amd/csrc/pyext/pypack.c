/* _pypack — host-side marshalling helper for the Python layer (not on the compute path).
 *
 * pack_le32(values, modulus) -> bytes: every Python int of the sequence `values`, reduced mod `modulus`, as a
 * 32-byte little-endian word.  This is what BatchProver.upload does per witness variable before handing the
 * buffer to plonk_prover_upload_variables; int.to_bytes + b"".join costs ~0.1 us per value in CPython and was
 * the bulk of the host time per proof.  Values already in [0, modulus) take the fast path (_PyLong_AsByteArray);
 * anything else (negative, >= modulus) goes through PyNumber_Remainder, i.e. Python's own `%`.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>

/* Fast path for CPython < 3.12 with 30-bit digits: a non-negative exact int of at most 9 digits is copied digit by digit
 * into four little-endian 64-bit words (_PyLong_AsByteArray walks it bit-accumulator style, ~50 ns; this is ~8 ns).
 * Returns 1 and fills dst when the value is in [0, modulus), 0 when the caller must take the general path. */
#if PY_VERSION_HEX < 0x030C0000 && PYLONG_BITS_IN_DIGIT == 30 && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
#define PYPACK_FAST_DIGITS 1
static int pack_fast(PyObject* v, const unsigned char* mod, unsigned char* dst) {
    if (!PyLong_CheckExact(v)) return 0;
    const Py_ssize_t sz = Py_SIZE(v);
    if (sz < 0 || sz > 9) return 0;
    const digit* d = ((PyLongObject*)v)->ob_digit;
    unsigned long long w[5] = {0, 0, 0, 0, 0};
    for (Py_ssize_t i = 0; i < sz; i++) {
        const unsigned bit = 30u * (unsigned)i, word = bit >> 6, sh = bit & 63u;
        w[word] |= (unsigned long long)d[i] << sh;
        if (sh > 34) w[word + 1] |= (unsigned long long)d[i] >> (64 - sh);
    }
    if (w[4]) return 0; /* >= 2^256 */
    unsigned long long m[4];
    memcpy(m, mod, 32);
    for (int i = 3; i >= 0; i--) {
        if (w[i] < m[i]) break;
        if (w[i] > m[i] || i == 0) return 0; /* >= modulus */
    }
    memcpy(dst, w, 32);
    return 1;
}
#else
#define PYPACK_FAST_DIGITS 0
static int pack_fast(PyObject* v, const unsigned char* mod, unsigned char* dst) { (void)v; (void)mod; (void)dst; return 0; }
#endif

static int below(const unsigned char* v, const unsigned char* m) { /* little-endian 32-byte compare: v < m */
    for (int i = 31; i >= 0; i--) {
        if (v[i] < m[i]) return 1;
        if (v[i] > m[i]) return 0;
    }
    return 0;
}

static PyObject* pack_le32(PyObject* self, PyObject* args) {
    PyObject *seq, *modulus;
    if (!PyArg_ParseTuple(args, "OO!", &seq, &PyLong_Type, &modulus)) return NULL;
    unsigned char mod[32];
    if (_PyLong_AsByteArray((PyLongObject*)modulus, mod, 32, 1, 0) < 0) return NULL;
    PyObject* fast = PySequence_Fast(seq, "pack_le32 expects a sequence of ints");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject* out = PyBytes_FromStringAndSize(NULL, 32 * n);
    if (!out) { Py_DECREF(fast); return NULL; }
    unsigned char* dst = (unsigned char*)PyBytes_AS_STRING(out);
    for (Py_ssize_t i = 0; i < n; i++, dst += 32) {
        PyObject* v = PySequence_Fast_GET_ITEM(fast, i); /* borrowed */
        if (pack_fast(v, mod, dst)) continue;
        PyObject* as_int = NULL;
        if (!PyLong_Check(v)) { /* objects with __int__ / __index__ (e.g. Scalar) */
            as_int = PyNumber_Long(v);
            if (!as_int) goto fail;
            v = as_int;
        }
        int ok = 0;
        if (Py_SIZE(v) >= 0 && _PyLong_NumBits(v) <= 256) {
            if (_PyLong_AsByteArray((PyLongObject*)v, dst, 32, 1, 0) == 0 && below(dst, mod)) ok = 1;
            else PyErr_Clear();
        }
        if (!ok) {
            PyObject* r = PyNumber_Remainder(v, modulus);
            if (!r) { Py_XDECREF(as_int); goto fail; }
            int rc = _PyLong_AsByteArray((PyLongObject*)r, dst, 32, 1, 0);
            Py_DECREF(r);
            if (rc < 0) { Py_XDECREF(as_int); goto fail; }
        }
        Py_XDECREF(as_int);
    }
    Py_DECREF(fast);
    return out;
fail:
    Py_DECREF(fast);
    Py_DECREF(out);
    return NULL;
}

/* one value -> 32 bytes at dst (reduced mod modulus); 0 on success */
static int pack_one(PyObject* v, PyObject* modulus, const unsigned char* mod, unsigned char* dst) {
    if (pack_fast(v, mod, dst)) return 0;
    PyObject* as_int = NULL;
    if (!PyLong_Check(v)) {
        as_int = PyNumber_Long(v);
        if (!as_int) return -1;
        v = as_int;
    }
    int ok = 0;
    if (Py_SIZE(v) >= 0 && _PyLong_NumBits(v) <= 256) {
        if (_PyLong_AsByteArray((PyLongObject*)v, dst, 32, 1, 0) == 0 && below(dst, mod)) ok = 1;
        else PyErr_Clear();
    }
    if (!ok) {
        PyObject* r = PyNumber_Remainder(v, modulus);
        if (!r) { Py_XDECREF(as_int); return -1; }
        int rc = _PyLong_AsByteArray((PyLongObject*)r, dst, 32, 1, 0);
        Py_DECREF(r);
        if (rc < 0) { Py_XDECREF(as_int); return -1; }
    }
    Py_XDECREF(as_int);
    return 0;
}

/* One witness by dictionary look-ups (the general path): 0 on success, -1 with an exception set. */
static int pack_row_by_lookup(PyObject* w, PyObject* kf, Py_ssize_t V, PyObject* modulus, const unsigned char* mod, unsigned char* row) {
    const int is_dict = PyDict_CheckExact(w);
    for (Py_ssize_t i = 0; i < V; i++) {
        PyObject* key = PySequence_Fast_GET_ITEM(kf, i);
        if (is_dict) {
            PyObject* v = PyDict_GetItemWithError(w, key); /* borrowed */
            if (!v) {
                if (!PyErr_Occurred()) PyErr_SetObject(PyExc_KeyError, key);
                return -1;
            }
            if (pack_one(v, modulus, mod, row + 32 * i) < 0) return -1;
        } else {
            PyObject* v = PyObject_GetItem(w, key); /* new reference */
            if (!v) return -1;
            const int rc = pack_one(v, modulus, mod, row + 32 * i);
            Py_DECREF(v);
            if (rc < 0) return -1;
        }
    }
    return 0;
}

/* pack_dicts_le32(witnesses, keys, modulus) -> bytes: for every dict of `witnesses`, the values of `keys` in order, each
 * reduced mod modulus, 32 bytes little-endian: the [B][V] buffer plonk_prover_upload_variables takes.  A missing key
 * raises KeyError(key), as witness[key] would.
 *
 * Witnesses of one circuit are normally built by the same code, so their dictionaries list the same key objects in the
 * same insertion order.  The first dictionary's order is resolved to column slots once (V look-ups); every dictionary is
 * then WALKED (PyDict_Next, ~5 ns per item) instead of probed V times (~25 ns each), checking each key against the
 * remembered one by identity, then equality.  A dictionary that deviates in any way — another size or order, a value
 * that is not an exact int in [0, modulus) — is redone by look-ups, so the result never depends on the shortcut. */
static PyObject* pack_dicts_le32(PyObject* self, PyObject* args) {
    PyObject *wits, *keys, *modulus;
    if (!PyArg_ParseTuple(args, "OOO!", &wits, &keys, &PyLong_Type, &modulus)) return NULL;
    unsigned char mod[32];
    if (_PyLong_AsByteArray((PyLongObject*)modulus, mod, 32, 1, 0) < 0) return NULL;
    PyObject* wf = PySequence_Fast(wits, "pack_dicts_le32 expects a sequence of dicts");
    if (!wf) return NULL;
    PyObject* kf = PySequence_Fast(keys, "pack_dicts_le32 expects a sequence of keys");
    if (!kf) { Py_DECREF(wf); return NULL; }
    const Py_ssize_t B = PySequence_Fast_GET_SIZE(wf), V = PySequence_Fast_GET_SIZE(kf);
    PyObject* out = PyBytes_FromStringAndSize(NULL, 32 * B * V);
    PyObject** okeys = NULL;   /* the first dictionary's keys in iteration order (borrowed: that dictionary stays alive in wf) */
    Py_ssize_t* oslot = NULL;  /* their column slots, -1 for a key the circuit does not use */
    Py_ssize_t n_items = -1;
    if (!out) goto fail;
    unsigned char* dst = (unsigned char*)PyBytes_AS_STRING(out);
    if (B > 1 && V > 0 && PYPACK_FAST_DIGITS && PyDict_CheckExact(PySequence_Fast_GET_ITEM(wf, 0))) {
        PyObject* first = PySequence_Fast_GET_ITEM(wf, 0);
        PyObject* slotmap = PyDict_New();
        n_items = PyDict_GET_SIZE(first);
        okeys = (PyObject**)PyMem_Malloc(sizeof(PyObject*) * (size_t)(n_items ? n_items : 1));
        oslot = (Py_ssize_t*)PyMem_Malloc(sizeof(Py_ssize_t) * (size_t)(n_items ? n_items : 1));
        int usable = slotmap && okeys && oslot;
        for (Py_ssize_t i = 0; usable && i < V; i++) {  /* key -> slot (a repeated key keeps the general path) */
            PyObject* key = PySequence_Fast_GET_ITEM(kf, i);
            PyObject* idx = PyLong_FromSsize_t(i);
            if (!idx || PyDict_Contains(slotmap, key) != 0 || PyDict_SetItem(slotmap, key, idx) < 0) usable = 0;
            Py_XDECREF(idx);
        }
        Py_ssize_t pos = 0, t = 0, covered = 0;
        PyObject *key, *value;
        while (usable && PyDict_Next(first, &pos, &key, &value)) {
            PyObject* idx = PyDict_GetItemWithError(slotmap, key); /* borrowed */
            if (!idx && PyErr_Occurred()) usable = 0;
            okeys[t] = key;
            oslot[t] = idx ? PyLong_AsSsize_t(idx) : -1;
            covered += idx != NULL;
            t++;
        }
        PyErr_Clear();
        Py_XDECREF(slotmap);
        if (!usable || covered != V) n_items = -1;  /* a key is missing: the general path raises the KeyError */
    }
    for (Py_ssize_t b = 0; b < B; b++, dst += 32 * V) {
        PyObject* w = PySequence_Fast_GET_ITEM(wf, b);
        int done = 0;
        if (n_items >= 0 && PyDict_CheckExact(w) && PyDict_GET_SIZE(w) == n_items) {
            Py_ssize_t pos = 0, t = 0;
            PyObject *key, *value;
            done = 1;
            while (PyDict_Next(w, &pos, &key, &value)) {  /* nothing below runs Python code: the walk is safe */
                if (key != okeys[t]) {
                    /* equal but distinct key objects: only exact str keys are compared here (no user __eq__) */
                    if (!(PyUnicode_CheckExact(key) && PyUnicode_CheckExact(okeys[t]) && PyUnicode_Compare(key, okeys[t]) == 0)) { done = 0; break; }
                }
                if (oslot[t] >= 0 && !pack_fast(value, mod, dst + 32 * oslot[t])) { done = 0; break; }
                t++;
            }
            if (PyErr_Occurred()) PyErr_Clear();
        }
        if (!done && pack_row_by_lookup(w, kf, V, modulus, mod, dst) < 0) goto fail;
    }
    PyMem_Free(okeys);
    PyMem_Free(oslot);
    Py_DECREF(wf);
    Py_DECREF(kf);
    return out;
fail:
    PyMem_Free(okeys);
    PyMem_Free(oslot);
    Py_DECREF(wf);
    Py_DECREF(kf);
    Py_XDECREF(out);
    return NULL;
}

static PyMethodDef methods[] = {{"pack_dicts_le32", pack_dicts_le32, METH_VARARGS, "dicts x keys -> [B][V] 32-byte little-endian words, reduced mod modulus"},
                                {"pack_le32", pack_le32, METH_VARARGS, "ints -> 32-byte little-endian words, reduced mod modulus"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pypack", "host-side int packing for plonkathon_amd", -1, methods};
PyMODINIT_FUNC PyInit__pypack(void) { return PyModule_Create(&moddef); }
