/* srcgather.c -- one C-level pass over a LIST of SrcParams objects (desi-mcmc_amd/celeste_src.py) into the arrays the C ABI takes.
 *
 * Host-side plumbing of the reference-API mirror, not part of the HIP path: celeste.list_cache("exact") -- the default -- re-reads
 * EVERY source of a list on EVERY call, as the reference does (CelestePy/celeste.py:203-219).  With numpy's C-level passes
 * (attrgetter + fromiter + concatenate) that costs 3.2 ms at 10 000 sources, 2.7 x the resident render; here the objects' slots
 * are read at their member offsets (SrcParams has __slots__) and the arrays' data pointers directly: ~0.3 ms.
 *
 * gather(srcs, offsets, band_letters, bidx, typ, radec, flux, shape) -> number of objects read on the fast path (== len(srcs) when
 * every object took it).  An object that does not fit the fast path -- not exactly a SrcParams, a temperature star (src.t), a
 * location or flux container that is not a C-contiguous float64 ndarray (or, fluxes, a dict of floats by band letter), a shape
 * attribute that is not a number -- stops the pass: the caller then gathers the whole list the general way
 * (celeste._gather_plain), so behaviour and error messages are the general path's.
 * No value is ever computed here beyond reading: counts are formed by the caller with the same numpy expression as before.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

enum { O_A, O_U, O_T, O_THETA, O_SIGMA, O_PHI, O_RHO, O_FLUXES, N_OFF };

static inline PyObject *slot(PyObject *o, Py_ssize_t off) { return *(PyObject **)((char *)o + off); }

static inline int f64_array(PyObject *o, npy_intp need, const double **data) {
    if (!o || !PyArray_CheckExact(o)) return 0;
    PyArrayObject *a = (PyArrayObject *)o;
    if (PyArray_TYPE(a) != NPY_FLOAT64 || PyArray_NDIM(a) != 1 || PyArray_DIM(a, 0) != need || !PyArray_IS_C_CONTIGUOUS(a) ||
        !PyArray_ISALIGNED(a) || PyArray_ISBYTESWAPPED(a))
        return 0;
    *data = (const double *)PyArray_DATA(a);
    return 1;
}

static inline int number(PyObject *o, double *v) {
    if (!o) return 0;
    if (PyFloat_Check(o)) { *v = PyFloat_AS_DOUBLE(o); return 1; }      /* float and numpy.float64 */
    if (PyLong_CheckExact(o)) { *v = PyLong_AsDouble(o); return !(*v == -1.0 && PyErr_Occurred()); }
    return 0;
}

static PyObject *gather(PyObject *self, PyObject *args) {
    PyObject *srcs, *cls, *offs, *letters, *bidx_o;
    PyArrayObject *typ, *radec, *flux, *shape, *untyped;
    if (!PyArg_ParseTuple(args, "O!OO!O!O!O!O!O!O!O!", &PyList_Type, &srcs, &cls, &PyTuple_Type, &offs, &PyTuple_Type, &letters,
                          &PyTuple_Type, &bidx_o, &PyArray_Type, &typ, &PyArray_Type, &radec, &PyArray_Type, &flux, &PyArray_Type, &shape,
                          &PyArray_Type, &untyped))
        return NULL;
    const Py_ssize_t S = PyList_GET_SIZE(srcs);
    const Py_ssize_t B = PyTuple_GET_SIZE(letters);
    if (PyTuple_GET_SIZE(offs) != N_OFF || PyTuple_GET_SIZE(bidx_o) != B || B < 1 || B > 16) {
        PyErr_SetString(PyExc_ValueError, "srcgather: bad offsets / bands");
        return NULL;
    }
    if (PyArray_TYPE(typ) != NPY_INT32 || PyArray_SIZE(typ) != S || !PyArray_IS_C_CONTIGUOUS(typ) ||
        PyArray_TYPE(radec) != NPY_FLOAT64 || PyArray_SIZE(radec) != 2 * S || !PyArray_IS_C_CONTIGUOUS(radec) ||
        PyArray_TYPE(flux) != NPY_FLOAT64 || PyArray_SIZE(flux) != B * S || !PyArray_IS_C_CONTIGUOUS(flux) ||
        PyArray_TYPE(shape) != NPY_FLOAT64 || PyArray_SIZE(shape) != 4 * S || !PyArray_IS_C_CONTIGUOUS(shape) ||
        PyArray_TYPE(untyped) != NPY_BOOL || PyArray_SIZE(untyped) != S || !PyArray_IS_C_CONTIGUOUS(untyped)) {
        PyErr_SetString(PyExc_ValueError, "srcgather: output arrays of the wrong type or size");
        return NULL;
    }
    Py_ssize_t off[N_OFF];
    for (int k = 0; k < N_OFF; k++) {
        off[k] = PyLong_AsSsize_t(PyTuple_GET_ITEM(offs, k));
        if (off[k] < (Py_ssize_t)sizeof(PyObject) || off[k] > 4096) { PyErr_SetString(PyExc_ValueError, "srcgather: bad slot offset"); return NULL; }
    }
    int bidx[16];
    for (Py_ssize_t b = 0; b < B; b++) {
        bidx[b] = (int)PyLong_AsLong(PyTuple_GET_ITEM(bidx_o, b));
        if (bidx[b] < 0 || bidx[b] > 4) { PyErr_SetString(PyExc_ValueError, "srcgather: band index outside ugriz"); return NULL; }
    }
    int32_t *ty = (int32_t *)PyArray_DATA(typ);
    double *rd = (double *)PyArray_DATA(radec), *fl = (double *)PyArray_DATA(flux), *sh = (double *)PyArray_DATA(shape);
    npy_bool *un = (npy_bool *)PyArray_DATA(untyped);
    Py_ssize_t s = 0;
    for (; s < S; s++) {
        PyObject *o = PyList_GET_ITEM(srcs, s);
        if ((PyObject *)Py_TYPE(o) != cls) break;
        PyObject *a = slot(o, off[O_A]), *t = slot(o, off[O_T]);
        if (t && t != Py_None) {                 /* a star given by temperature: the photometry hook's path */
            int truth = PyObject_IsTrue(t);
            if (truth != 0) { if (truth < 0) PyErr_Clear(); break; }
        }
        int is_gal = 0;
        if (!a) break;
        if (a == Py_None) un[s] = 1;
        else if (PyLong_Check(a) || PyIndex_Check(a)) {          /* int, bool, numpy integer scalars */
            Py_ssize_t av = PyLong_CheckExact(a) ? (Py_ssize_t)PyLong_AsLong(a) : PyNumber_AsSsize_t(a, NULL);
            if (av == -1 && PyErr_Occurred()) { PyErr_Clear(); break; }
            if (av != 0 && av != 1) break;       /* (whatever else it is: the general path's business) */
            is_gal = (int)av; un[s] = 0;
        } else break;
        ty[s] = is_gal;
        const double *ud;
        if (!f64_array(slot(o, off[O_U]), 2, &ud)) break;
        rd[2 * s] = ud[0]; rd[2 * s + 1] = ud[1];
        double *shs = sh + 4 * s;
        shs[0] = shs[1] = shs[2] = shs[3] = 0.0;
        if (is_gal) {
            if (!number(slot(o, off[O_THETA]), shs + 0) || !number(slot(o, off[O_SIGMA]), shs + 1) ||
                !number(slot(o, off[O_PHI]), shs + 2) || !number(slot(o, off[O_RHO]), shs + 3)) { PyErr_Clear(); break; }
        }
        PyObject *f = slot(o, off[O_FLUXES]);
        const double *fd;
        if (f64_array(f, 5, &fd)) {
            for (Py_ssize_t b = 0; b < B; b++) fl[s * B + b] = fd[bidx[b]];
        } else if (f && PyDict_CheckExact(f)) {
            Py_ssize_t b = 0;
            for (; b < B; b++) {
                PyObject *v = PyDict_GetItemWithError(f, PyTuple_GET_ITEM(letters, b));      /* borrowed */
                if (!v || !number(v, fl + s * B + b)) { PyErr_Clear(); break; }
            }
            if (b < B) break;
        } else break;
    }
    return PyLong_FromSsize_t(s);
}

/* member offsets of the slots named in `names` on class `cls` (a class with __slots__: its attributes are member descriptors) */
static PyObject *slot_offsets(PyObject *self, PyObject *args) {
    PyObject *cls, *names;
    if (!PyArg_ParseTuple(args, "OO!", &cls, &PyTuple_Type, &names)) return NULL;
    if (!PyType_Check(cls)) { PyErr_SetString(PyExc_TypeError, "slot_offsets: a class"); return NULL; }
    const Py_ssize_t n = PyTuple_GET_SIZE(names);
    PyObject *out = PyTuple_New(n);
    if (!out) return NULL;
    for (Py_ssize_t k = 0; k < n; k++) {
        PyObject *d = PyObject_GetAttr(cls, PyTuple_GET_ITEM(names, k));
        if (!d) { Py_DECREF(out); return NULL; }
        if (Py_TYPE(d) != &PyMemberDescr_Type || ((PyMemberDescrObject *)d)->d_member->type != T_OBJECT_EX) {
            Py_DECREF(d); Py_DECREF(out);
            PyErr_SetString(PyExc_TypeError, "slot_offsets: not a __slots__ member");
            return NULL;
        }
        PyTuple_SET_ITEM(out, k, PyLong_FromSsize_t(((PyMemberDescrObject *)d)->d_member->offset));
        Py_DECREF(d);
    }
    return out;
}

static PyMethodDef methods[] = {
    {"gather", gather, METH_VARARGS, "one pass over a list of SrcParams into typ / radec / flux / shape / untyped arrays -> objects read"},
    {"slot_offsets", slot_offsets, METH_VARARGS, "member offsets of __slots__ attributes"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_srcgather", "C-level gather of SrcParams lists (host plumbing)", -1, methods};
PyMODINIT_FUNC PyInit__srcgather(void) {
    import_array();
    return PyModule_Create(&moddef);
}
