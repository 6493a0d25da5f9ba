/* Host-side helper of the class API (CPython C API, no GPU): the Python objects the reference's tracking loop iterates
 * over, built in C.
 *
 * `matches` is a list of one-element lists [[cv2.DMatch], ...] in the reference (src/v2/frame.py:33-47), and main.py
 * walks it per frame (`[point_ids[m[0].queryIdx] for m in matches]`, main.py:187-188,210).  With ~420 matches per frame that
 * is ~850 objects per frame; created from Python (tuple subclass + property(itemgetter)) they cost 130 us per frame of the
 * class-API period -- more than the GPU's PnP.  Here: a DMatch type with C member descriptors and `rows()`, which builds
 * the whole list of lists from the three int32 arrays in one call.  frame.py falls back to its Python classes if this
 * module was not built. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <structmember.h>

typedef struct {
  PyObject_HEAD
  int queryIdx, trainIdx, imgIdx;
  double distance;
} DMatchObject;

static PyTypeObject DMatchType;

static PyObject* dmatch_new(PyTypeObject* type, PyObject* args, PyObject* kw) {
  static char* names[] = {"queryIdx", "trainIdx", "distance", "imgIdx", NULL};
  int q = 0, t = 0, img = 0;
  double d = 0.0;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "|iidi", names, &q, &t, &d, &img)) return NULL;
  DMatchObject* m = (DMatchObject*)type->tp_alloc(type, 0);
  if (!m) return NULL;
  m->queryIdx = q;
  m->trainIdx = t;
  m->imgIdx = img;
  m->distance = d;
  return (PyObject*)m;
}

static PyObject* dmatch_repr(DMatchObject* m) {
  char buf[128];
  PyOS_snprintf(buf, sizeof buf, "DMatch(queryIdx=%d, trainIdx=%d, distance=%g)", m->queryIdx, m->trainIdx, m->distance);
  return PyUnicode_FromString(buf);
}

static PyMemberDef dmatch_members[] = {
    {"queryIdx", T_INT, offsetof(DMatchObject, queryIdx), 0, "index into the query (first) descriptor set"},
    {"trainIdx", T_INT, offsetof(DMatchObject, trainIdx), 0, "index into the train (second) descriptor set"},
    {"imgIdx", T_INT, offsetof(DMatchObject, imgIdx), 0, "train image index (always 0 here)"},
    {"distance", T_DOUBLE, offsetof(DMatchObject, distance), 0, "Hamming distance"},
    {NULL, 0, 0, 0, NULL}};

static PyTypeObject DMatchType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "visual_slam_amd._rows.DMatch",
    .tp_basicsize = sizeof(DMatchObject),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_doc = "The fields of cv2.DMatch the reference reads: queryIdx, trainIdx, imgIdx, distance.",
    .tp_new = dmatch_new,
    .tp_repr = (reprfunc)dmatch_repr,
    .tp_members = dmatch_members,
};

/* rows(query_idx, train_idx, distance) -> [[DMatch], ...]; the arguments are C-contiguous int32 buffers of one length */
static PyObject* rows(PyObject* self, PyObject* args) {
  Py_buffer bq, bt, bd;
  if (!PyArg_ParseTuple(args, "y*y*y*", &bq, &bt, &bd)) return NULL;
  PyObject* out = NULL;
  if (bq.len != bt.len || bq.len != bd.len || bq.len % 4 != 0) {
    PyErr_SetString(PyExc_ValueError, "rows: three int32 buffers of one length expected");
    goto done;
  }
  {
    const Py_ssize_t n = bq.len / 4;
    const int32_t* q = (const int32_t*)bq.buf;
    const int32_t* t = (const int32_t*)bt.buf;
    const int32_t* d = (const int32_t*)bd.buf;
    out = PyList_New(n);
    if (!out) goto done;
    for (Py_ssize_t i = 0; i < n; ++i) {
      DMatchObject* m = PyObject_New(DMatchObject, &DMatchType);
      PyObject* inner = m ? PyList_New(1) : NULL;
      if (!inner) {
        Py_XDECREF((PyObject*)m);
        Py_CLEAR(out);
        goto done;
      }
      m->queryIdx = q[i];
      m->trainIdx = t[i];
      m->imgIdx = 0;
      m->distance = (double)d[i];
      PyList_SET_ITEM(inner, 0, (PyObject*)m);
      PyList_SET_ITEM(out, i, inner);
    }
  }
done:
  PyBuffer_Release(&bq);
  PyBuffer_Release(&bt);
  PyBuffer_Release(&bd);
  return out;
}

/* pack(seq, out) -> True / False.  seq: a list (or tuple) of n buffers -- the per-point position / image point / descriptor rows
 * the reference's callers hand over ONE BY ONE (Point(location=pt, ...), AddFrame(uv=..., descriptor=...), main.py:130-135,
 * 312-318); out: a writable C-contiguous buffer of n rows.  Every item must be a C-contiguous buffer of exactly out's row
 * length in bytes and of out's item size and format; then row i of out receives item i (one memcpy each) and the result is True.
 * Anything else -- another dtype, a strided view, a ragged row -- returns False with out partly written: the caller converts the
 * slow way.  (np.array(list_of_595_small_arrays) costs 100 us; this is 10.) */
static PyObject* pack(PyObject* self, PyObject* args) {
  PyObject* seq;
  Py_buffer out;
  if (!PyArg_ParseTuple(args, "Ow*", &seq, &out)) return NULL;
  PyObject* fast = PySequence_Fast(seq, "pack: a sequence of buffers");
  if (!fast) {
    PyBuffer_Release(&out);
    return NULL;
  }
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  int ok = PyBuffer_IsContiguous(&out, 'C') && n > 0 && out.len % n == 0;
  const Py_ssize_t row = ok ? out.len / n : 0;
  PyObject** items = PySequence_Fast_ITEMS(fast);
  for (Py_ssize_t i = 0; ok && i < n; ++i) {
    Py_buffer b;
    if (PyObject_GetBuffer(items[i], &b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
      PyErr_Clear();
      ok = 0;
      break;
    }
    if (b.len != row || b.itemsize != out.itemsize || (b.format && out.format && strcmp(b.format, out.format) != 0))
      ok = 0;
    else
      memcpy((char*)out.buf + i * row, b.buf, (size_t)row);
    PyBuffer_Release(&b);
  }
  Py_DECREF(fast);
  PyBuffer_Release(&out);
  if (ok) Py_RETURN_TRUE;
  Py_RETURN_FALSE;
}

/* collect(points) -> (locs, rev_sum, counts, fids, frame_objs, uvs, descs) or None.  points: a sequence of Point objects (point.py:
 * slots `_loc`, `_rev`, `_frames` = {frame id: (Frame, uv, descriptor)}).  One pass in C over what Map._absorb_added otherwise reads
 * out of the objects with Python-level loops: locs[k] = the position object of point k, rev_sum = the sum of the `_rev` counters,
 * counts[k] = its number of observations, and -- point by point, in dict order -- the frame id, Frame object, image point and
 * descriptor of every observation.  None when an object does not look like a Point (the caller then walks the objects itself). */
static PyObject *s_loc, *s_rev, *s_frames;
static PyObject* collect(PyObject* self, PyObject* arg) {
  PyObject* fast = PySequence_Fast(arg, "collect: a sequence of points");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  PyObject** items = PySequence_Fast_ITEMS(fast);
  PyObject *locs = PyList_New(n), *counts = PyList_New(n), *fids = PyList_New(0), *fobjs = PyList_New(0), *uvs = PyList_New(0),
           *descs = PyList_New(0), *out = NULL;
  long long rev_sum = 0;
  int ok = locs && counts && fids && fobjs && uvs && descs;
  Py_ssize_t filled = 0;
  for (Py_ssize_t k = 0; ok && k < n; ++k) {
    PyObject* loc = PyObject_GetAttr(items[k], s_loc);
    PyObject* rev = loc ? PyObject_GetAttr(items[k], s_rev) : NULL;
    PyObject* fr = rev ? PyObject_GetAttr(items[k], s_frames) : NULL;
    if (!fr || !PyDict_Check(fr) || !PyLong_Check(rev)) {
      Py_XDECREF(loc);
      Py_XDECREF(rev);
      Py_XDECREF(fr);
      ok = 0;
      break;
    }
    rev_sum += PyLong_AsLongLong(rev);
    Py_DECREF(rev);
    PyList_SET_ITEM(locs, k, loc);  /* steals */
    PyObject* cnt = PyLong_FromSsize_t(PyDict_GET_SIZE(fr));
    if (!cnt) {
      Py_DECREF(fr);
      ok = 0;
      break;
    }
    PyList_SET_ITEM(counts, k, cnt);
    filled = k + 1;
    Py_ssize_t pos = 0;
    PyObject *key, *val;
    while (ok && PyDict_Next(fr, &pos, &key, &val)) {
      if (!PyTuple_Check(val) || PyTuple_GET_SIZE(val) != 3 || PyList_Append(fids, key) || PyList_Append(fobjs, PyTuple_GET_ITEM(val, 0)) ||
          PyList_Append(uvs, PyTuple_GET_ITEM(val, 1)) || PyList_Append(descs, PyTuple_GET_ITEM(val, 2)))
        ok = 0;
    }
    Py_DECREF(fr);
  }
  /* (PyList_New leaves NULL items: a list that was not filled to the end must not be handed out or traversed) */
  if (locs && filled < n) {
    for (Py_ssize_t k = filled; k < n; ++k) {
      if (!PyList_GET_ITEM(locs, k)) {
        Py_INCREF(Py_None);
        PyList_SET_ITEM(locs, k, Py_None);
      }
      if (counts && !PyList_GET_ITEM(counts, k)) {
        Py_INCREF(Py_None);
        PyList_SET_ITEM(counts, k, Py_None);
      }
    }
  }
  if (ok) out = Py_BuildValue("(OLOOOOO)", locs, rev_sum, counts, fids, fobjs, uvs, descs);
  Py_XDECREF(locs);
  Py_XDECREF(counts);
  Py_XDECREF(fids);
  Py_XDECREF(fobjs);
  Py_XDECREF(uvs);
  Py_XDECREF(descs);
  Py_DECREF(fast);
  if (!ok) {
    if (PyErr_Occurred()) {
      if (PyErr_ExceptionMatches(PyExc_MemoryError)) return NULL;
      PyErr_Clear();
    }
    Py_RETURN_NONE;
  }
  return out;
}

static PyMethodDef methods[] = {{"rows", rows, METH_VARARGS, "rows(query_idx, train_idx, distance: int32 buffers) -> [[DMatch], ...]"},
                                {"pack", pack, METH_VARARGS, "pack(seq of equal buffers, out) -> bool: item i copied into row i of out"},
                                {"collect", collect, METH_O, "collect(points) -> (locs, rev_sum, counts, fids, frame_objs, uvs, descs) or None"},
                                {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_rows", "match rows of the class API, built in C (no GPU)", -1, methods};

PyMODINIT_FUNC PyInit__rows(void) {
  if (PyType_Ready(&DMatchType) < 0) return NULL;
  s_loc = PyUnicode_InternFromString("_loc");
  s_rev = PyUnicode_InternFromString("_rev");
  s_frames = PyUnicode_InternFromString("_frames");
  if (!s_loc || !s_rev || !s_frames) return NULL;
  PyObject* m = PyModule_Create(&module);
  if (!m) return NULL;
  Py_INCREF(&DMatchType);
  if (PyModule_AddObject(m, "DMatch", (PyObject*)&DMatchType) < 0) {
    Py_DECREF(&DMatchType);
    Py_DECREF(m);
    return NULL;
  }
  return m;
}
