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

static PyMethodDef methods[] = {{"pack_le32", pack_le32, METH_VARARGS, "ints -> 32-byte little-endian words, reduced mod modulus"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pypack", "host-side int packing for plonkathon_amd", -1, methods};
PyMODINIT_FUNC PyInit__pypack(void) { return PyModule_Create(&moddef); }
