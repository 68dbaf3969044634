/* _mirfast: the three C-ABI calls that sit between two launches of GenesisEnv.step (mir_step_prepare, mir_step_go, mir_step_end,
 * include/mirigid.h) as CPython built-ins.
 *
 * Why: env.step is bound by the host's turn-around between the moment a launch's terminated bytes arrive and the moment the next
 * launch is in the queue (tools/probes/host_turnaround.py: every microsecond there is a microsecond per step).  A ctypes foreign
 * call costs ~0.39 us of argument conversion per call, a METH_FASTCALL built-in ~0.06 us.  Nothing is computed here: the functions
 * take addresses as Python ints and pass them on.  The library's functions are bound by address (bind(), from the ctypes handle
 * that gym_genesis/backend/lib.py already holds), so this module does not link against libmirigid.so and there is still exactly one
 * copy of the library in the process.  Plain C, no torch, no HIP. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*prepare_fn)(void*, float*, float*, float*, uint8_t*);
typedef int (*go_fn)(void*, const float*, void*);
typedef int (*end_fn)(void*, uint8_t*);
static prepare_fn f_prepare;
static go_fn f_go;
static end_fn f_end;

static int as_ptr(PyObject* o, void** out) {
  if (o == Py_None) { *out = NULL; return 0; }
  *out = PyLong_AsVoidPtr(o);
  return (*out == NULL && PyErr_Occurred()) ? -1 : 0;
}

/* bind(prepare_addr, go_addr, end_addr): addresses of mir_step_prepare / mir_step_go / mir_step_end in the loaded library */
static PyObject* py_bind(PyObject* self, PyObject* const* args, Py_ssize_t n) {
  void *p, *g, *e;
  if (n != 3) { PyErr_SetString(PyExc_TypeError, "bind(prepare_addr, go_addr, end_addr)"); return NULL; }
  if (as_ptr(args[0], &p) || as_ptr(args[1], &g) || as_ptr(args[2], &e)) return NULL;
  if (!p || !g || !e) { PyErr_SetString(PyExc_ValueError, "bind: null function address"); return NULL; }
  f_prepare = (prepare_fn)p; f_go = (go_fn)g; f_end = (end_fn)e;
  Py_RETURN_NONE;
}

/* prepare(handle, agent_pos, env_state, reward, terminated) -> rc */
static PyObject* py_prepare(PyObject* self, PyObject* const* args, Py_ssize_t n) {
  void* v[5];
  if (n != 5 || !f_prepare) { PyErr_SetString(PyExc_TypeError, "prepare(handle, agent_pos, env_state, reward, terminated) after bind()"); return NULL; }
  for (int i = 0; i < 5; i++)
    if (as_ptr(args[i], &v[i])) return NULL;
  return PyLong_FromLong(f_prepare(v[0], (float*)v[1], (float*)v[2], (float*)v[3], (uint8_t*)v[4]));
}

/* go(handle, action, stream) -> rc */
static PyObject* py_go(PyObject* self, PyObject* const* args, Py_ssize_t n) {
  void* v[3];
  if (n != 3 || !f_go) { PyErr_SetString(PyExc_TypeError, "go(handle, action, stream) after bind()"); return NULL; }
  for (int i = 0; i < 3; i++)
    if (as_ptr(args[i], &v[i])) return NULL;
  return PyLong_FromLong(f_go(v[0], (const float*)v[1], v[2]));
}

/* end(handle, terminated_host) -> rc.  Waits (tens of microseconds at most) for the launch's bytes; the interpreter lock is kept:
 * giving it up and taking it back costs more than the wait is worth to another thread. */
static PyObject* py_end(PyObject* self, PyObject* const* args, Py_ssize_t n) {
  void* v[2];
  if (n != 2 || !f_end) { PyErr_SetString(PyExc_TypeError, "end(handle, terminated_host) after bind()"); return NULL; }
  for (int i = 0; i < 2; i++)
    if (as_ptr(args[i], &v[i])) return NULL;
  return PyLong_FromLong(f_end(v[0], (uint8_t*)v[1]));
}

static PyMethodDef methods[] = {
    {"bind", (PyCFunction)(void (*)(void))py_bind, METH_FASTCALL, "bind(prepare_addr, go_addr, end_addr)"},
    {"prepare", (PyCFunction)(void (*)(void))py_prepare, METH_FASTCALL, "mir_step_prepare"},
    {"go", (PyCFunction)(void (*)(void))py_go, METH_FASTCALL, "mir_step_go"},
    {"end", (PyCFunction)(void (*)(void))py_end, METH_FASTCALL, "mir_step_end"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_mirfast", "built-in versions of the three calls between two env.step launches", -1, methods};
PyMODINIT_FUNC PyInit__mirfast(void) { return PyModule_Create(&moddef); }
