/* pack_module.c -- CPython helper: Python byte tokens -> one packed buffer + int64 offsets.
 *
 * The host side of MinHash.update_batch / bulk with the default hashfunc (ref: datasketch/minhash.py:262-263
 * hashes one Python object at a time; here the tokens of a whole batch go to the device as a CSR of
 * bytes, see sha1_kernels.hip).  Walking 10^6 small `bytes` objects with map(len) and b"".join costs
 * ~180 ns per token in the interpreter; this loop costs ~10.  Host glue only: no device code, no
 * numpy C API (results are bytearrays that numpy wraps without copying).
 *
 *   pack_tokens(tokens)      -> (data: bytearray, byte_offsets: bytearray of int64[n+1])
 *   pack_sets(sets)          -> (data, byte_offsets int64[T+1], set_offsets int64[N+1])
 *
 * A token must be `bytes` or support the buffer protocol with one-byte items; anything else raises
 * the TypeError hashlib.sha1 would raise for it.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* length of a token; -1 with an exception set */
static Py_ssize_t token_size(PyObject *tok) {
    if (PyBytes_CheckExact(tok)) return PyBytes_GET_SIZE(tok);
    if (PyByteArray_CheckExact(tok)) return PyByteArray_GET_SIZE(tok);
    if (PyUnicode_Check(tok)) {
        PyErr_SetString(PyExc_TypeError, "Strings must be encoded before hashing");
        return -1;
    }
    Py_buffer view;
    if (PyObject_GetBuffer(tok, &view, PyBUF_SIMPLE) != 0) {
        PyErr_Clear();
        PyErr_Format(PyExc_TypeError, "object supporting the buffer API required, got %.200s", Py_TYPE(tok)->tp_name);
        return -1;
    }
    const Py_ssize_t n = view.len;
    PyBuffer_Release(&view);
    return n;
}

/* copy a token's bytes to dst (size known from token_size); -1 with an exception set */
static int token_copy(PyObject *tok, char *dst, Py_ssize_t expect) {
    if (PyBytes_CheckExact(tok)) {
        if (PyBytes_GET_SIZE(tok) != expect) goto changed;
        memcpy(dst, PyBytes_AS_STRING(tok), (size_t)expect);
        return 0;
    }
    if (PyByteArray_CheckExact(tok)) {
        if (PyByteArray_GET_SIZE(tok) != expect) goto changed;
        memcpy(dst, PyByteArray_AS_STRING(tok), (size_t)expect);
        return 0;
    }
    {
        Py_buffer view;
        if (PyObject_GetBuffer(tok, &view, PyBUF_SIMPLE) != 0) return -1;
        if (view.len != expect) {
            PyBuffer_Release(&view);
            goto changed;
        }
        memcpy(dst, view.buf, (size_t)expect);
        PyBuffer_Release(&view);
        return 0;
    }
changed:
    PyErr_SetString(PyExc_RuntimeError, "a token changed size while it was being packed");
    return -1;
}

static PyObject *new_offsets(Py_ssize_t count) {  /* zero-filled int64[count] as a bytearray */
    PyObject *arr = PyByteArray_FromStringAndSize(NULL, count * (Py_ssize_t)sizeof(int64_t));
    if (arr) memset(PyByteArray_AS_STRING(arr), 0, (size_t)count * sizeof(int64_t));
    return arr;
}

/* offsets[0..n] of the tokens of one fast sequence, starting at byte position *pos; -1 on error */
static int measure(PyObject *fast, int64_t *offsets, int64_t *pos) {
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject **items = PySequence_Fast_ITEMS(fast);
    for (Py_ssize_t i = 0; i < n; ++i) {
        const Py_ssize_t len = token_size(items[i]);
        if (len < 0) return -1;
        offsets[i] = *pos;
        *pos += len;
    }
    return 0;
}

static int fill(PyObject *fast, const int64_t *offsets, int64_t end, char *data) {
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject **items = PySequence_Fast_ITEMS(fast);
    for (Py_ssize_t i = 0; i < n; ++i) {
        const int64_t next = i + 1 < n ? offsets[i + 1] : end;
        if (token_copy(items[i], data + offsets[i], (Py_ssize_t)(next - offsets[i])) != 0) return -1;
    }
    return 0;
}

static PyObject *pack_tokens(PyObject *self, PyObject *arg) {
    (void)self;
    PyObject *fast = PySequence_Fast(arg, "tokens must be an iterable of bytes-like objects");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject *offs = new_offsets(n + 1), *data = NULL, *result = NULL;
    if (!offs) goto done;
    {
        int64_t *offsets = (int64_t *)PyByteArray_AS_STRING(offs);
        int64_t pos = 0;
        if (measure(fast, offsets, &pos) != 0) goto done;
        offsets[n] = pos;
        data = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)pos);
        if (!data) goto done;
        if (fill(fast, offsets, pos, PyByteArray_AS_STRING(data)) != 0) goto done;
        result = PyTuple_Pack(2, data, offs);
    }
done:
    Py_XDECREF(data);
    Py_XDECREF(offs);
    Py_DECREF(fast);
    return result;
}

static PyObject *pack_sets(PyObject *self, PyObject *arg) {
    (void)self;
    PyObject *outer = PySequence_Fast(arg, "sets must be an iterable of iterables of bytes-like objects");
    if (!outer) return NULL;
    const Py_ssize_t n_sets = PySequence_Fast_GET_SIZE(outer);
    PyObject **sets = PySequence_Fast_ITEMS(outer);
    PyObject *inner = PyList_New(n_sets);  /* the fast form of every set, kept alive between the two passes */
    PyObject *set_offs = NULL, *tok_offs = NULL, *data = NULL, *result = NULL;
    if (!inner) goto done;
    set_offs = new_offsets(n_sets + 1);
    if (!set_offs) goto done;
    {
        int64_t *set_offsets = (int64_t *)PyByteArray_AS_STRING(set_offs);
        int64_t n_tokens = 0;
        for (Py_ssize_t s = 0; s < n_sets; ++s) {
            PyObject *fast = PySequence_Fast(sets[s], "every set must be an iterable of bytes-like objects");
            if (!fast) goto done;
            PyList_SET_ITEM(inner, s, fast);
            set_offsets[s] = n_tokens;
            n_tokens += PySequence_Fast_GET_SIZE(fast);
        }
        set_offsets[n_sets] = n_tokens;
        tok_offs = new_offsets((Py_ssize_t)n_tokens + 1);
        if (!tok_offs) goto done;
        int64_t *tok_offsets = (int64_t *)PyByteArray_AS_STRING(tok_offs);
        int64_t pos = 0;
        for (Py_ssize_t s = 0; s < n_sets; ++s)
            if (measure(PyList_GET_ITEM(inner, s), tok_offsets + set_offsets[s], &pos) != 0) goto done;
        tok_offsets[n_tokens] = pos;
        data = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)pos);
        if (!data) goto done;
        char *bytes = PyByteArray_AS_STRING(data);
        for (Py_ssize_t s = 0; s < n_sets; ++s)
            if (fill(PyList_GET_ITEM(inner, s), tok_offsets + set_offsets[s], tok_offsets[set_offsets[s + 1]], bytes) != 0)
                goto done;
        result = PyTuple_Pack(3, data, tok_offs, set_offs);
    }
done:
    Py_XDECREF(data);
    Py_XDECREF(tok_offs);
    Py_XDECREF(set_offs);
    Py_XDECREF(inner);
    Py_DECREF(outer);
    return result;
}

/* Sets (lists / tuples) of exact Python ints -> (uint64 values, int64 set offsets), what numpy's
 * np.array(tokens, dtype=uint64) makes of every set (ref: datasketch/minhash.py:294), at ~10 ns per token instead of ~90.
 * None when anything else turns up (numpy arrays, numpy scalars, floats, other iterables): the caller then takes the
 * numpy route, with numpy's own conversions and errors.  OverflowError for values outside uint64, as numpy raises. */
static PyObject *pack_int_sets(PyObject *self, PyObject *arg) {
    (void)self;
    if (!PyList_CheckExact(arg) && !PyTuple_CheckExact(arg)) Py_RETURN_NONE;
    const Py_ssize_t n_sets = PySequence_Fast_GET_SIZE(arg);
    PyObject **sets = PySequence_Fast_ITEMS(arg);
    int64_t n_tokens = 0;
    for (Py_ssize_t s = 0; s < n_sets; ++s) {
        if (!PyList_CheckExact(sets[s]) && !PyTuple_CheckExact(sets[s])) Py_RETURN_NONE;
        n_tokens += PySequence_Fast_GET_SIZE(sets[s]);
    }
    PyObject *set_offs = new_offsets(n_sets + 1), *vals = NULL, *result = NULL;
    if (!set_offs) return NULL;
    vals = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)n_tokens * (Py_ssize_t)sizeof(uint64_t));
    if (!vals) goto done;
    {
        int64_t *set_offsets = (int64_t *)PyByteArray_AS_STRING(set_offs);
        uint64_t *out = (uint64_t *)PyByteArray_AS_STRING(vals);
        int64_t pos = 0;
        for (Py_ssize_t s = 0; s < n_sets; ++s) {
            const Py_ssize_t n = PySequence_Fast_GET_SIZE(sets[s]);
            PyObject **items = PySequence_Fast_ITEMS(sets[s]);
            set_offsets[s] = pos;
            if (pos + n > n_tokens) {  /* a set grew under our feet */
                PyErr_SetString(PyExc_RuntimeError, "a set changed size while it was being packed");
                goto done;
            }
            for (Py_ssize_t i = 0; i < n; ++i) {
                if (!PyLong_CheckExact(items[i])) {  /* bool, numpy scalar, float ...: numpy decides */
                    result = Py_None;
                    Py_INCREF(result);
                    goto done;
                }
                const unsigned long long v = PyLong_AsUnsignedLongLong(items[i]);
                if (v == (unsigned long long)-1 && PyErr_Occurred()) goto done;  /* OverflowError: negative or >= 2^64 */
                out[pos++] = (uint64_t)v;
            }
        }
        set_offsets[n_sets] = pos;
        if (pos != n_tokens) {
            PyErr_SetString(PyExc_RuntimeError, "a set changed size while it was being packed");
            goto done;
        }
        result = PyTuple_Pack(2, vals, set_offs);
    }
done:
    Py_XDECREF(vals);
    Py_XDECREF(set_offs);
    return result;
}

static PyMethodDef methods[] = {
    {"pack_int_sets", pack_int_sets, METH_O, "lists of Python ints -> (uint64 values bytearray, int64 set offsets bytearray) or None"},
    {"pack_tokens", pack_tokens, METH_O, "tokens -> (data bytearray, int64 byte offsets bytearray)"},
    {"pack_sets", pack_sets, METH_O, "sets of tokens -> (data, int64 byte offsets, int64 set offsets)"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_mhxpack", "pack Python byte tokens for libmhx", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__mhxpack(void) { return PyModule_Create(&module); }
