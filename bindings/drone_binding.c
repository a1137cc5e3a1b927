/*
 * drone_binding.c — the CPython extension a PufferLib-side `binding.c` for this env
 * would be: vec_init / vec_reset / vec_step / vec_log / vec_close over the caller's
 * shared buffers, forwarding to the C-ABI of include/drone_vec.h (one HIP launch per
 * vec call) instead of looping a per-env c_step on the CPU.
 *
 * Reference interface replaced: PufferLib ocean envs include a generic
 * `env_binding.h` that defines these module functions around `c_step` [UNVERIFIED
 * RECOLLECTION — the reference snapshot has no binding source to cite:
 * /root/reference/.gitmodules:1-3 names an empty `pufferlib` submodule]. Names,
 * argument order (five buffers, num_envs, seed, env kwargs) and the dict returned
 * by vec_log follow that convention as far as the north-star describes it.
 *
 * Pure CPython C-API + the buffer protocol: no numpy headers, no HIP headers.
 *   gcc -O2 -shared -fPIC $(python3-config --includes) -Iinclude bindings/drone_binding.c \
 *       -Ldrone_amd -l:libdrone_hip.so -Wl,-rpath,'$ORIGIN' -o drone_amd/drone_binding$(python3-config --extension-suffix)
 *
 * Buffers: any object exporting a writable C-contiguous buffer (numpy arrays, slices
 * of a shared-memory block) -> DRONE_BUFFERS_HOST; or objects with `data_ptr()` (torch
 * tensors on the GPU) -> DRONE_BUFFERS_DEVICE, zero-copy; or any DLPack producer on a
 * ROCm device (`__dlpack__`: cupy / jax arrays, torch tensors behind a wrapper) ->
 * DRONE_BUFFERS_DEVICE as well; or five times None -> the library allocates the buffers
 * in HBM and vec_dlpack(handle, name) hands them out as DLPack capsules
 * (torch.from_dlpack / cupy.from_dlpack), SURVEY.md §8 f3.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stddef.h>
#include <string.h>

#include "drone_vec.h"

#define CAPSULE_NAME "drone_amd.DroneVec"

/* ---- DLPack (the data-interchange ABI of dmlc/dlpack, v0.x unversioned structs): declared here so that the module
 * needs no header beyond Python.h. Field order and widths are the standard's. ---- */
enum { kDLCPU = 1, kDLROCM = 10 };
enum { kDLInt = 0, kDLUInt = 1, kDLFloat = 2, kDLBool = 6 };
typedef struct DLDevice { int32_t device_type; int32_t device_id; } DLDevice;
typedef struct DLDataType { uint8_t code; uint8_t bits; uint16_t lanes; } DLDataType;
typedef struct DLTensor {
    void* data;
    DLDevice device;
    int32_t ndim;
    DLDataType dtype;
    int64_t* shape;
    int64_t* strides; /* in elements; NULL = compact row-major */
    uint64_t byte_offset;
} DLTensor;
typedef struct DLManagedTensor {
    DLTensor dl_tensor;
    void* manager_ctx;
    void (*deleter)(struct DLManagedTensor* self);
} DLManagedTensor;
#define DLPACK_CAPSULE "dltensor"
#define DLPACK_CAPSULE_USED "used_dltensor"

typedef struct Handle {
    DroneVec* v;
    Py_buffer views[5]; /* host mode: keeps the exporters' memory pinned while the env uses it */
    int n_views;
    PyObject* keep[5];  /* device mode: strong references to the tensors */
    int n_keep;
    DLManagedTensor* imported[5]; /* device mode, DLPack producers: the managed tensors this env consumed (released at close) */
    int n_imported;
    size_t n, obs_dim;  /* envs, floats per observation row */
    int device_kind;    /* 1: buffers are device tensors */
    int exports;        /* DLPack views of the env's buffers handed out by vec_dlpack and not yet deleted */
} Handle;

static void release_buffers(Handle* h) {
    for (int i = 0; i < h->n_views; i++) PyBuffer_Release(&h->views[i]);
    h->n_views = 0;
    for (int i = 0; i < h->n_keep; i++) Py_XDECREF(h->keep[i]);
    h->n_keep = 0;
    for (int i = 0; i < h->n_imported; i++)
        if (h->imported[i] && h->imported[i]->deleter) h->imported[i]->deleter(h->imported[i]);
    h->n_imported = 0;
}

static void handle_free(Handle* h) {
    if (!h) return;
    if (h->v) drone_vec_close(h->v);
    release_buffers(h);
    PyMem_Free(h);
}

static void capsule_destructor(PyObject* cap) {
    Handle* h = (Handle*)PyCapsule_GetPointer(cap, CAPSULE_NAME);
    if (!h) { PyErr_Clear(); return; }
    handle_free(h);
}

static Handle* get_handle(PyObject* cap) {
    Handle* h = (Handle*)PyCapsule_GetPointer(cap, CAPSULE_NAME);
    if (!h) return NULL;
    if (!h->v) { PyErr_SetString(PyExc_ValueError, "env handle is closed"); return NULL; }
    return h;
}

static int raise_if_failed(Handle* h) {
    if (drone_vec_status(h->v)) {
        PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_vec_status_message(h->v));
        return -1;
    }
    return 0;
}

/* ---- kwargs -> DroneConfig ---- */
typedef struct Field { const char* name; char kind; size_t off; } Field; /* kind: 'i' int32, 'u' uint32, 'f' float */
#define F_(name, kind) {#name, kind, offsetof(DroneConfig, name)}
static const Field kFields[] = {
    F_(task, 'i'), F_(device, 'i'), F_(env_offset, 'u'), F_(horizon, 'i'), F_(substeps, 'i'), F_(compact_done, 'i'), F_(agents_per_env, 'i'),
    F_(dt, 'f'), F_(mass, 'f'), F_(arm, 'f'), F_(ixx, 'f'), F_(iyy, 'f'), F_(izz, 'f'),
    F_(k_thrust, 'f'), F_(k_torque, 'f'), F_(k_drag, 'f'), F_(k_ang_damp, 'f'), F_(gravity, 'f'),
    F_(max_rpm, 'f'), F_(motor_tau, 'f'), F_(max_vel, 'f'), F_(max_omega, 'f'),
    F_(bound, 'f'), F_(spawn_extent, 'f'), F_(target_extent, 'f'), F_(tilt_init, 'f'),
    F_(hover_radius, 'f'), F_(waypoint_radius, 'f'), F_(wind_theta, 'f'), F_(wind_sigma, 'f'), F_(wind_max, 'f'),
    F_(c_omega, 'f'), F_(c_action, 'f'), F_(crash_penalty, 'f'), F_(progress_scale, 'f'), F_(waypoint_bonus, 'f'),
    F_(collision_radius, 'f'), F_(proximity_radius, 'f'), F_(c_proximity, 'f'), F_(gate_radius, 'f'), F_(host_pages_exclusive, 'i'), F_(state_layout, 'i'),
};
#undef F_

static int apply_kwargs(DroneConfig* cfg, PyObject* kwargs) {
    if (!kwargs) return 0;
    PyObject *key, *val;
    Py_ssize_t pos = 0;
    while (PyDict_Next(kwargs, &pos, &key, &val)) {
        const char* k = PyUnicode_AsUTF8(key);
        if (!k) return -1;
        if (!strcmp(k, "task")) continue; /* consumed before drone_config_default */
        const Field* f = NULL;
        for (size_t i = 0; i < sizeof(kFields) / sizeof(kFields[0]); i++)
            if (!strcmp(k, kFields[i].name)) { f = &kFields[i]; break; }
        if (!f) { PyErr_Format(PyExc_TypeError, "vec_init: unknown env kwarg '%s'", k); return -1; }
        char* dst = (char*)cfg + f->off;
        if (f->kind == 'f') {
            const double d = PyFloat_AsDouble(val);
            if (d == -1.0 && PyErr_Occurred()) return -1;
            *(float*)dst = (float)d;
        } else {
            const long long x = PyLong_AsLongLong(val);
            if (x == -1 && PyErr_Occurred()) return -1;
            if (f->kind == 'u') *(uint32_t*)dst = (uint32_t)x;
            else *(int32_t*)dst = (int32_t)x;
        }
    }
    return 0;
}

/* a device tensor: has data_ptr(); returns 1 and the address, 0 if not that kind, -1 on error.
 * 2 (no address yet): no data_ptr() but a DLPack producer — see dl_import. */
static int device_pointer(PyObject* o, void** out) {
    if (!PyObject_HasAttrString(o, "data_ptr")) return PyObject_HasAttrString(o, "__dlpack__") && !PyObject_CheckBuffer(o) ? 2 : 0;
    PyObject* r = PyObject_CallMethod(o, "data_ptr", NULL);
    if (!r) return -1;
    *out = PyLong_AsVoidPtr(r);
    Py_DECREF(r);
    if (PyErr_Occurred()) return -1;
    return 1;
}

/* Consume `o.__dlpack__()`: a compact row-major ROCm tensor of f32 (want_float) or one-byte items holding at least
 * need_bytes. On success the managed tensor is OURS (*out; release with its deleter), *ptr its first element and
 * *device its HIP ordinal. */
static int dl_import(PyObject* o, int want_float, size_t need_bytes, const char* name, DLManagedTensor** out, void** ptr, int* device) {
    PyObject* cap = PyObject_CallMethod(o, "__dlpack__", NULL);
    if (!cap) return -1;
    DLManagedTensor* m = (DLManagedTensor*)PyCapsule_GetPointer(cap, DLPACK_CAPSULE);
    if (!m) { Py_DECREF(cap); return -1; }
    const DLTensor* t = &m->dl_tensor;
    const char* why = NULL;
    int type_error = 0;
    size_t items = 1;
    for (int d = 0; d < t->ndim; d++) items *= (size_t)t->shape[d];
    if (t->device.device_type != kDLROCM) why = "is not on a ROCm device";
    else if (t->dtype.lanes != 1 || (want_float ? !(t->dtype.code == kDLFloat && t->dtype.bits == 32)
                                               : !((t->dtype.code == kDLUInt || t->dtype.code == kDLInt || t->dtype.code == kDLBool) && t->dtype.bits == 8)))
        why = want_float ? "must hold float32 items" : "must hold uint8 / bool items", type_error = 1;
    else if (items * (want_float ? 4u : 1u) < need_bytes) why = "is too small";
    if (!why && t->strides) { /* compact row-major, ignoring extent-1 dimensions */
        int64_t expect = 1;
        for (int d = t->ndim - 1; d >= 0; d--) {
            if (t->shape[d] != 1 && t->strides[d] != expect) { why = "must be contiguous"; break; }
            expect *= t->shape[d];
        }
    }
    if (why) {
        PyErr_Format(type_error ? PyExc_TypeError : PyExc_ValueError, "%s (DLPack): %s", name, why);
        Py_DECREF(cap); /* still named "dltensor": the producer's capsule destructor releases it */
        return -1;
    }
    if (PyCapsule_SetName(cap, DLPACK_CAPSULE_USED) != 0) { Py_DECREF(cap); return -1; }
    Py_DECREF(cap);
    *out = m;
    *ptr = (char*)t->data + t->byte_offset;
    *device = t->device.device_id;
    return 0;
}

/* Host buffers must hold what the env writes through them: f32 items for observations / actions / rewards, one-byte
 * items for the flags (a float64 or int32 array of sufficient byte length would otherwise be written through a wrong
 * layout silently). `want_float`: 1 = 4-byte float items, 0 = 1-byte items. The view was requested with PyBUF_FORMAT. */
static int check_host_format(const Py_buffer* view, int want_float, const char* name) {
    const char* f = view->format ? view->format : "B";
    while (*f == '@' || *f == '=' || *f == '<') f++; /* native / little-endian prefixes */
    const int ok = want_float ? (view->itemsize == 4 && f[0] == 'f' && f[1] == 0)
                              : (view->itemsize == 1 && (f[0] == 'B' || f[0] == 'b' || f[0] == '?' || f[0] == 'c') && f[1] == 0);
    if (!ok) {
        PyErr_Format(PyExc_TypeError, "%s: expected %s items, got format '%s' (itemsize %zd)", name, want_float ? "float32" : "uint8 / bool", view->format ? view->format : "B", view->itemsize);
        return -1;
    }
    return 0;
}

/* Device tensors: dtype by element_size() + is_floating_point(), layout by is_contiguous(), size by numel(). */
static int check_device_tensor(PyObject* t, int want_float, size_t need_bytes, const char* name) {
    PyObject* ne = PyObject_CallMethod(t, "numel", NULL);
    PyObject* es = ne ? PyObject_CallMethod(t, "element_size", NULL) : NULL;
    if (!ne || !es) { Py_XDECREF(ne); Py_XDECREF(es); return -1; }
    const size_t esz = PyLong_AsSize_t(es), have = PyLong_AsSize_t(ne) * esz;
    Py_DECREF(ne);
    Py_DECREF(es);
    if (PyErr_Occurred()) return -1;
    if (have < need_bytes) { PyErr_Format(PyExc_ValueError, "%s holds %zu bytes, %zu needed", name, have, need_bytes); return -1; }
    if (esz != (want_float ? 4u : 1u)) { PyErr_Format(PyExc_TypeError, "%s: expected %s elements, got %zu-byte ones", name, want_float ? "float32" : "uint8 / bool", esz); return -1; }
    if (PyObject_HasAttrString(t, "is_floating_point")) {
        PyObject* fp = PyObject_CallMethod(t, "is_floating_point", NULL);
        if (!fp) return -1;
        const int is_fp = PyObject_IsTrue(fp);
        Py_DECREF(fp);
        if (is_fp != (want_float ? 1 : 0)) { PyErr_Format(PyExc_TypeError, "%s: expected a %s tensor", name, want_float ? "float32" : "uint8 / bool"); return -1; }
    }
    if (PyObject_HasAttrString(t, "is_contiguous")) {
        PyObject* c = PyObject_CallMethod(t, "is_contiguous", NULL);
        if (!c) return -1;
        const int contiguous = PyObject_IsTrue(c);
        Py_DECREF(c);
        if (!contiguous) { PyErr_Format(PyExc_ValueError, "%s must be contiguous", name); return -1; }
    }
    return 0;
}

/* vec_init(observations, actions, rewards, terminals, truncations, num_envs, seed, **env_kwargs) -> handle */
static PyObject* vec_init(PyObject* self, PyObject* args, PyObject* kwargs) {
    (void)self;
    PyObject* bufs[5];
    int num_envs;
    unsigned long long seed;
    if (!PyArg_ParseTuple(args, "OOOOOiK", &bufs[0], &bufs[1], &bufs[2], &bufs[3], &bufs[4], &num_envs, &seed)) return NULL;
    if (num_envs <= 0) { PyErr_SetString(PyExc_ValueError, "num_envs must be positive"); return NULL; }

    int task = DRONE_TASK_HOVER;
    if (kwargs) {
        PyObject* t = PyDict_GetItemString(kwargs, "task");
        if (t) {
            task = (int)PyLong_AsLong(t);
            if (task == -1 && PyErr_Occurred()) return NULL;
        }
    }
    DroneConfig cfg;
    drone_config_default(&cfg, task);
    if (apply_kwargs(&cfg, kwargs) < 0) return NULL;

    Handle* h = (Handle*)PyMem_Calloc(1, sizeof(Handle));
    if (!h) return PyErr_NoMemory();
    void* ptr[5] = {0};
    const size_t od = (size_t)drone_obs_dim(task);
    const size_t need[5] = {(size_t)num_envs * od * 4, (size_t)num_envs * DRONE_ACT_DIM * 4, (size_t)num_envs * 4, (size_t)num_envs, (size_t)num_envs};
    static const char* names[5] = {"observations", "actions", "rewards", "terminals", "truncations"};
    int kind = -1; /* 0 host, 1 device */
    int n_none = 0;
    for (int i = 0; i < 5; i++) n_none += bufs[i] == Py_None;
    if (n_none && n_none != 5) {
        PyErr_SetString(PyExc_TypeError, "vec_init: pass all five buffers, or None five times for library-owned device buffers");
        handle_free(h);
        return NULL;
    }
    const int have_device_kwarg = kwargs && PyDict_GetItemString(kwargs, "device") != NULL;
    for (int i = 0; i < 5 && !n_none; i++) {
        void* d = NULL;
        int isdev = device_pointer(bufs[i], &d);
        if (isdev < 0) { handle_free(h); return NULL; }
        if (isdev == 2) { /* a DLPack producer: the env consumes its managed tensor and keeps it until close */
            if (kind == 0) { PyErr_SetString(PyExc_TypeError, "vec_init: buffers must be all host buffers or all device tensors"); handle_free(h); return NULL; }
            int dl_dev = 0;
            if (dl_import(bufs[i], i < 3, need[i], names[i], &h->imported[h->n_imported], &d, &dl_dev) < 0) { handle_free(h); return NULL; }
            h->n_imported++;
            if (!have_device_kwarg && h->n_imported == 1) cfg.device = dl_dev;
            if (dl_dev != cfg.device) {
                PyErr_Format(PyExc_ValueError, "vec_init: %s lives on ROCm device %d, the env on device %d", names[i], dl_dev, cfg.device);
                handle_free(h);
                return NULL;
            }
            kind = 1;
            ptr[i] = d;
            continue;
        }
        if (kind >= 0 && kind != isdev) {
            PyErr_SetString(PyExc_TypeError, "vec_init: buffers must be all host buffers or all device tensors");
            handle_free(h);
            return NULL;
        }
        kind = isdev;
        if (isdev) {
            if (check_device_tensor(bufs[i], i < 3, need[i], names[i]) < 0) { handle_free(h); return NULL; }
            Py_INCREF(bufs[i]);
            h->keep[h->n_keep++] = bufs[i];
            ptr[i] = d;
        } else {
            if (PyObject_GetBuffer(bufs[i], &h->views[h->n_views], PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) { handle_free(h); return NULL; }
            h->n_views++;
            if (check_host_format(&h->views[i], i < 3, names[i]) < 0) { handle_free(h); return NULL; }
            if ((size_t)h->views[i].len < need[i]) {
                PyErr_Format(PyExc_ValueError, "vec_init: %s holds %zd bytes, %zu needed", names[i], h->views[i].len, need[i]);
                handle_free(h);
                return NULL;
            }
            ptr[i] = h->views[i].buf;
        }
    }
    if (n_none) kind = 1; /* ptr[] stays NULL: drone_vec_init allocates */
    cfg.buffer_kind = kind ? DRONE_BUFFERS_DEVICE : DRONE_BUFFERS_HOST;
    h->n = (size_t)num_envs;
    h->obs_dim = od;
    h->device_kind = kind;
    Py_BEGIN_ALLOW_THREADS
    h->v = drone_vec_init((float*)ptr[0], (float*)ptr[1], (float*)ptr[2], (unsigned char*)ptr[3], (unsigned char*)ptr[4], num_envs, seed, &cfg);
    Py_END_ALLOW_THREADS
    if (!h->v) {
        PyErr_Format(PyExc_RuntimeError, "drone_vec_init failed: %s", drone_last_error());
        handle_free(h);
        return NULL;
    }
    PyObject* cap = PyCapsule_New(h, CAPSULE_NAME, capsule_destructor);
    if (!cap) handle_free(h);
    return cap;
}

static PyObject* vec_reset(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    unsigned long long seed = 0;
    if (!PyArg_ParseTuple(args, "O|K", &cap, &seed)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    Py_BEGIN_ALLOW_THREADS
    drone_vec_reset(h->v, seed);
    Py_END_ALLOW_THREADS
    if (raise_if_failed(h) < 0) return NULL;
    Py_RETURN_NONE;
}

static PyObject* vec_step(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    Py_BEGIN_ALLOW_THREADS
    drone_vec_step(h->v);
    Py_END_ALLOW_THREADS
    if (raise_if_failed(h) < 0) return NULL;
    Py_RETURN_NONE;
}

/* vec_send(handle) / vec_recv(handle): vec_step in two halves (drone_vec_step_send / drone_vec_step_recv) */
static PyObject* vec_send(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    Py_BEGIN_ALLOW_THREADS
    drone_vec_step_send(h->v);
    Py_END_ALLOW_THREADS
    if (raise_if_failed(h) < 0) return NULL;
    Py_RETURN_NONE;
}

static PyObject* vec_recv(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    Py_BEGIN_ALLOW_THREADS
    drone_vec_step_recv(h->v);
    Py_END_ALLOW_THREADS
    if (raise_if_failed(h) < 0) return NULL;
    Py_RETURN_NONE;
}

/* vec_rollout(handle, horizon): fused rollout under the device-side random policy (SPEC.md §9) */
static PyObject* vec_rollout(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    int horizon;
    if (!PyArg_ParseTuple(args, "Oi", &cap, &horizon)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    Py_BEGIN_ALLOW_THREADS
    drone_vec_rollout(h->v, horizon);
    Py_END_ALLOW_THREADS
    if (raise_if_failed(h) < 0) return NULL;
    Py_RETURN_NONE;
}

/* vec_step_many(handle, k_steps, actions | None, observations, rewards, terminals, truncations): K env steps in one
 * launch with every step's outputs, into K-major blocks of the env's buffer kind ([K][N][4] in, [K][N][O] / [K][N] out);
 * actions = None draws the SPEC.md random policy in the kernel. */
static PyObject* step_many_impl(PyObject* args, int repeat) {
    PyObject *cap, *blk[5];
    int k_steps;
    if (!PyArg_ParseTuple(args, "OiOOOOO", &cap, &k_steps, &blk[0], &blk[1], &blk[2], &blk[3], &blk[4])) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    if (k_steps < 1) { PyErr_SetString(PyExc_ValueError, "vec_step_many: k_steps must be positive"); return NULL; }
    const size_t K = (size_t)k_steps;
    const size_t need[5] = {(repeat ? 1 : K) * h->n * DRONE_ACT_DIM * 4, K * h->n * h->obs_dim * 4, K * h->n * 4, K * h->n, K * h->n};
    if (repeat && blk[0] == Py_None) { PyErr_SetString(PyExc_ValueError, "vec_step_repeat: actions is None (the in-kernel policy is vec_step_many with actions = None)"); return NULL; }
    static const char* names[5] = {"actions", "observations", "rewards", "terminals", "truncations"};
    void* ptr[5] = {0};
    Py_buffer views[5];
    DLManagedTensor* lent[5]; /* DLPack producers: held for the duration of the call */
    int n_views = 0, n_lent = 0, ok = 1;
    for (int i = 0; i < 5 && ok; i++) {
        if (i == 0 && blk[0] == Py_None) continue; /* device policy */
        void* d = NULL;
        int isdev = device_pointer(blk[i], &d);
        if (isdev < 0) { ok = 0; break; }
        if (isdev == 2 && h->device_kind) {
            int dl_dev = 0;
            if (dl_import(blk[i], i < 3, need[i], names[i], &lent[n_lent], &d, &dl_dev) < 0) { ok = 0; break; }
            n_lent++;
            if (dl_dev != drone_vec_device(h->v)) { PyErr_Format(PyExc_ValueError, "vec_step_many: %s lives on ROCm device %d, the env on device %d", names[i], dl_dev, drone_vec_device(h->v)); ok = 0; break; }
            ptr[i] = d;
            continue;
        }
        if (isdev != h->device_kind) { PyErr_Format(PyExc_TypeError, "vec_step_many: %s must be of the env's buffer kind (host buffer / device tensor)", names[i]); ok = 0; break; }
        if (isdev) {
            if (check_device_tensor(blk[i], i < 3, need[i], names[i]) < 0) { ok = 0; break; }
            ptr[i] = d;
        } else {
            if (PyObject_GetBuffer(blk[i], &views[n_views], (i == 0 ? 0 : PyBUF_WRITABLE) | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) { ok = 0; break; }
            n_views++;
            if ((size_t)views[n_views - 1].len < need[i]) { PyErr_Format(PyExc_ValueError, "vec_step_many: %s holds %zd bytes, %zu needed", names[i], views[n_views - 1].len, need[i]); ok = 0; break; }
            if (check_host_format(&views[n_views - 1], i < 3, names[i]) < 0) { ok = 0; break; }
            ptr[i] = views[n_views - 1].buf;
        }
    }
    if (ok) {
        Py_BEGIN_ALLOW_THREADS
        if (repeat) drone_vec_step_repeat(h->v, k_steps, (const float*)ptr[0], (float*)ptr[1], (float*)ptr[2], (unsigned char*)ptr[3], (unsigned char*)ptr[4]);
        else drone_vec_step_many(h->v, k_steps, (const float*)ptr[0], (float*)ptr[1], (float*)ptr[2], (unsigned char*)ptr[3], (unsigned char*)ptr[4]);
        Py_END_ALLOW_THREADS
    }
    for (int i = 0; i < n_views; i++) PyBuffer_Release(&views[i]);
    for (int i = 0; i < n_lent; i++)
        if (lent[i]->deleter) lent[i]->deleter(lent[i]); /* the caller's object keeps the memory, as with data_ptr() tensors */
    if (!ok || raise_if_failed(h) < 0) return NULL;
    Py_RETURN_NONE;
}

static PyObject* vec_step_many(PyObject* self, PyObject* args) {
    (void)self;
    return step_many_impl(args, 0);
}

/* vec_step_repeat(handle, k_steps, actions [N][4], observations, rewards, terminals, truncations): action repeat / frame skip */
static PyObject* vec_step_repeat(PyObject* self, PyObject* args) {
    (void)self;
    return step_many_impl(args, 1);
}

/* vec_done_list_at(handle, k) -> bytes of uint32 ids (compact_done=1): envs that finished in step k of the last vec_step_many */
static PyObject* vec_done_list_at(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    int k;
    if (!PyArg_ParseTuple(args, "Oi", &cap, &k)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    uint32_t* ids = (uint32_t*)PyMem_Malloc(sizeof(uint32_t) * (h->n ? h->n : 1));
    if (!ids) return PyErr_NoMemory();
    const int cnt = drone_vec_done_list_at(h->v, k, ids, (int)h->n);
    if (cnt < 0) { PyMem_Free(ids); PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_last_error()); return NULL; }
    PyObject* out = PyBytes_FromStringAndSize((const char*)ids, (Py_ssize_t)sizeof(uint32_t) * cnt);
    PyMem_Free(ids);
    return out;
}

static PyObject* vec_log(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    DroneLog l;
    Py_BEGIN_ALLOW_THREADS
    drone_vec_log(h->v, &l);
    Py_END_ALLOW_THREADS
    if (raise_if_failed(h) < 0) return NULL;
    return Py_BuildValue("{s:f,s:f,s:f,s:f,s:f,s:f}", "perf", l.perf, "score", l.score, "episode_return", l.episode_return,
                         "episode_length", l.episode_length, "oob", l.oob, "n", l.n);
}

static PyObject* vec_close(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = (Handle*)PyCapsule_GetPointer(cap, CAPSULE_NAME);
    if (!h) return NULL;
    if (h->exports > 0) { /* tensors made from vec_dlpack capsules still point into the env's buffers */
        PyErr_Format(PyExc_RuntimeError, "vec_close: %d DLPack view(s) of this env's buffers are still alive; delete them first", h->exports);
        return NULL;
    }
    if (h->v) {
        drone_vec_close(h->v);
        h->v = NULL;
    }
    release_buffers(h);
    Py_RETURN_NONE;
}

/* ---- DLPack export ---- */
typedef struct Export {
    DLManagedTensor m; /* first: the deleter gets &m */
    int64_t shape[2];
    PyObject* owner;   /* the env handle capsule: the buffers live as long as it does */
    Handle* h;
} Export;

static void export_deleter(DLManagedTensor* m) {
    Export* e = (Export*)m;
    if (Py_IsInitialized()) { /* consumers may drop the tensor from any thread, without the GIL */
        PyGILState_STATE g = PyGILState_Ensure();
        e->h->exports--;
        Py_DECREF(e->owner);
        PyGILState_Release(g);
    }
    free(e);
}

static void export_capsule_destructor(PyObject* cap) { /* never consumed: the capsule still owns the managed tensor */
    if (!PyCapsule_IsValid(cap, DLPACK_CAPSULE)) return; /* renamed "used_dltensor": the consumer owns it */
    DLManagedTensor* m = (DLManagedTensor*)PyCapsule_GetPointer(cap, DLPACK_CAPSULE);
    if (m && m->deleter) m->deleter(m);
}

/* vec_dlpack(handle, name) -> DLPack capsule of one of the env's DEVICE buffers as bound now: "observations" [N][O] f32,
 * "actions" [N][4] f32, "rewards" [N] f32, "terminals" / "truncations" [N] u8 — zero-copy, on ROCm device
 * drone_vec_device(). torch.from_dlpack(capsule) / cupy.from_dlpack wrap it; the env (and with it the memory) stays
 * alive until every such tensor is gone, and vec_close refuses to run before that. */
static PyObject* vec_dlpack(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    const char* name;
    if (!PyArg_ParseTuple(args, "Os", &cap, &name)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    if (!h->device_kind) { PyErr_SetString(PyExc_TypeError, "vec_dlpack: this env uses host buffers (the caller's own arrays)"); return NULL; }
    float *obs, *act, *rew;
    unsigned char *term, *trunc;
    if (drone_vec_buffers(h->v, &obs, &act, &rew, &term, &trunc) != 0) { PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_last_error()); return NULL; }
    Export* e = (Export*)calloc(1, sizeof(Export));
    if (!e) return PyErr_NoMemory();
    DLTensor* t = &e->m.dl_tensor;
    t->shape = e->shape;
    t->shape[0] = (int64_t)h->n;
    t->ndim = 1;
    t->dtype.lanes = 1;
    t->dtype.code = kDLFloat;
    t->dtype.bits = 32;
    if (!strcmp(name, "observations")) { t->data = obs; t->ndim = 2; t->shape[1] = (int64_t)h->obs_dim; }
    else if (!strcmp(name, "actions")) { t->data = act; t->ndim = 2; t->shape[1] = DRONE_ACT_DIM; }
    else if (!strcmp(name, "rewards")) t->data = rew;
    else if (!strcmp(name, "terminals") || !strcmp(name, "truncations")) { t->data = name[1] == 'e' ? term : trunc; t->dtype.code = kDLUInt; t->dtype.bits = 8; }
    else { free(e); PyErr_Format(PyExc_ValueError, "vec_dlpack: unknown buffer '%s'", name); return NULL; }
    t->device.device_type = kDLROCM;
    t->device.device_id = drone_vec_device(h->v);
    e->m.deleter = export_deleter;
    e->m.manager_ctx = e;
    e->h = h;
    PyObject* out = PyCapsule_New(&e->m, DLPACK_CAPSULE, export_capsule_destructor);
    if (!out) { free(e); return NULL; }
    Py_INCREF(cap);
    e->owner = cap;
    h->exports++;
    return out;
}

/* vec_set_stream(handle, hip_stream_address): device-buffer mode launches on this stream from now on */
static PyObject* vec_set_stream(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    unsigned long long s;
    if (!PyArg_ParseTuple(args, "OK", &cap, &s)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    if (drone_vec_set_stream(h->v, (void*)(uintptr_t)s) != 0) { PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_last_error()); return NULL; }
    Py_RETURN_NONE;
}

/* vec_fill_random_actions(handle, actions=None, gstep=None): the SPEC.md §2 random policy into the action buffer */
static PyObject* vec_fill_random_actions(PyObject* self, PyObject* args) {
    (void)self;
    PyObject *cap, *buf = Py_None, *gs = Py_None;
    if (!PyArg_ParseTuple(args, "O|OO", &cap, &buf, &gs)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    uint32_t g = drone_vec_gstep(h->v);
    if (gs != Py_None) {
        g = (uint32_t)PyLong_AsUnsignedLongMask(gs);
        if (PyErr_Occurred()) return NULL;
    }
    void* p = NULL;
    Py_buffer view;
    int have_view = 0;
    DLManagedTensor* lent = NULL;
    if (buf == Py_None) { /* the env's bound action buffer, whoever owns it */
        float* a = NULL;
        if (drone_vec_buffers(h->v, NULL, &a, NULL, NULL, NULL) != 0) { PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_last_error()); return NULL; }
        p = a;
    } else {
        int isdev = device_pointer(buf, &p);
        if (isdev < 0) return NULL;
        const size_t need = h->n * DRONE_ACT_DIM * 4;  /* the library writes this many bytes through the pointer */
        if (isdev == 2 && h->device_kind) {
            int dl_dev = 0;
            if (dl_import(buf, 1, need, "actions", &lent, &p, &dl_dev) < 0) return NULL;
            if (dl_dev != drone_vec_device(h->v)) { lent->deleter(lent); PyErr_SetString(PyExc_ValueError, "vec_fill_random_actions: actions lives on another device"); return NULL; }
        } else if (isdev != h->device_kind) { PyErr_SetString(PyExc_TypeError, "vec_fill_random_actions: the buffer must be of the env's buffer kind (host buffer / device tensor)"); return NULL; }
        else if (isdev) {
            if (check_device_tensor(buf, 1, need, "actions") < 0) return NULL;
        } else {
            if (PyObject_GetBuffer(buf, &view, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) return NULL;
            have_view = 1;
            if ((size_t)view.len < need) { PyErr_Format(PyExc_ValueError, "vec_fill_random_actions: actions holds %zd bytes, %zu needed", view.len, need); PyBuffer_Release(&view); return NULL; }
            if (check_host_format(&view, 1, "actions") < 0) { PyBuffer_Release(&view); return NULL; }
            p = view.buf;
        }
    }
    const int rc = drone_vec_fill_random_actions(h->v, (float*)p, g);
    if (have_view) PyBuffer_Release(&view);
    if (lent && lent->deleter) lent->deleter(lent);
    if (rc != 0) { PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_last_error()); return NULL; }
    Py_RETURN_NONE;
}

static PyObject* vec_gstep(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    return PyLong_FromUnsignedLong(drone_vec_gstep(h->v));
}

/* vec_device(handle) -> HIP device ordinal of the env; vec_sync(handle): wait for the env's stream */
static PyObject* vec_device(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    return PyLong_FromLong(drone_vec_device(h->v));
}

/* vec_host_pin(handle, buffer, pages_exclusive) / vec_host_unpin(handle, buffer): pin a page-owning host block (any object
 * with a writable C-contiguous buffer over a mapping of its own: a numpy array over its own mmap, a shared-memory block) so
 * that vec_step_many / vec_step_repeat access it in place. pages_exclusive = 1 is the caller's word that the block is such
 * a mapping (round 5: a page-aligned block inside the malloc heap must not be registered, and the library cannot tell);
 * left at its default 0 the call fails with that explanation unless the block is already pinned. The caller keeps the
 * object alive and unpins it before letting it go. */
static PyObject* host_pin_impl(PyObject* args, int pin) {
    PyObject *cap, *buf;
    int excl = 0;
    if (!PyArg_ParseTuple(args, pin ? "OO|i" : "OO", &cap, &buf, &excl)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    Py_buffer view;
    if (PyObject_GetBuffer(buf, &view, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) < 0) return NULL;
    const int rc = pin ? drone_vec_host_pin(h->v, view.buf, (size_t)view.len, excl) : drone_vec_host_unpin(h->v, view.buf);
    PyBuffer_Release(&view);
    if (rc != 0) { PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_last_error()); drone_vec_clear_status(h->v); return NULL; }
    Py_RETURN_NONE;
}
static PyObject* vec_host_pin(PyObject* self, PyObject* args) { (void)self; return host_pin_impl(args, 1); }
static PyObject* vec_host_unpin(PyObject* self, PyObject* args) { (void)self; return host_pin_impl(args, 0); }

/* vec_host_transport(handle) -> 0 mirror, 1 zero-copy, 2 zero-copy through pinned stand-ins, 3 the same moved by the host copy pool, -1 device buffers */
static PyObject* vec_variant(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    return PyUnicode_FromString(drone_vec_variant(h->v));
}

static PyObject* vec_host_transport(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    return PyLong_FromLong(drone_vec_host_transport(h->v));
}

static PyObject* vec_sync(PyObject* self, PyObject* args) {
    (void)self;
    PyObject* cap;
    if (!PyArg_ParseTuple(args, "O", &cap)) return NULL;
    Handle* h = get_handle(cap);
    if (!h) return NULL;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = drone_vec_sync(h->v);
    Py_END_ALLOW_THREADS
    if (rc != 0) { PyErr_Format(PyExc_RuntimeError, "libdrone_hip: %s", drone_last_error()); return NULL; }
    Py_RETURN_NONE;
}

static PyObject* obs_dim(PyObject* self, PyObject* args) {
    (void)self;
    int task;
    if (!PyArg_ParseTuple(args, "i", &task)) return NULL;
    return PyLong_FromLong(drone_obs_dim(task));
}

static PyMethodDef methods[] = {
    {"vec_init", (PyCFunction)(void (*)(void))vec_init, METH_VARARGS | METH_KEYWORDS,
     "vec_init(observations, actions, rewards, terminals, truncations, num_envs, seed, **env_kwargs) -> handle"},
    {"vec_reset", vec_reset, METH_VARARGS, "vec_reset(handle, seed=0)"},
    {"vec_step", vec_step, METH_VARARGS, "vec_step(handle): read actions, advance every env, overwrite the output buffers"},
    {"vec_send", vec_send, METH_VARARGS, "vec_send(handle): first half of vec_step — read the actions, enqueue the step, return without waiting"},
    {"vec_recv", vec_recv, METH_VARARGS, "vec_recv(handle): second half — wait for the sent step, outputs are in the buffers"},
    {"vec_rollout", vec_rollout, METH_VARARGS, "vec_rollout(handle, horizon): fused rollout under the device-side random policy"},
    {"vec_step_many", vec_step_many, METH_VARARGS,
     "vec_step_many(handle, k_steps, actions | None, observations, rewards, terminals, truncations): K env steps in one launch, every step's outputs in K-major blocks"},
    {"vec_step_repeat", vec_step_repeat, METH_VARARGS,
     "vec_step_repeat(handle, k_steps, actions [N][4], observations, rewards, terminals, truncations): K env steps under one action block (frame skip), every step's outputs in K-major blocks"},
    {"vec_done_list_at", vec_done_list_at, METH_VARARGS, "vec_done_list_at(handle, k) -> bytes (uint32 ids) of the envs that finished in step k of the last vec_step_many"},
    {"vec_log", vec_log, METH_VARARGS, "vec_log(handle) -> dict(perf, score, episode_return, episode_length, oob, n)"},
    {"vec_close", vec_close, METH_VARARGS, "vec_close(handle)"},
    {"vec_dlpack", vec_dlpack, METH_VARARGS, "vec_dlpack(handle, name) -> DLPack capsule (ROCm device) of the env's observations / actions / rewards / terminals / truncations"},
    {"vec_set_stream", vec_set_stream, METH_VARARGS, "vec_set_stream(handle, hip_stream_address)"},
    {"vec_fill_random_actions", vec_fill_random_actions, METH_VARARGS, "vec_fill_random_actions(handle, actions=None, gstep=None)"},
    {"vec_gstep", vec_gstep, METH_VARARGS, "vec_gstep(handle) -> int"},
    {"vec_device", vec_device, METH_VARARGS, "vec_device(handle) -> HIP device ordinal the env lives on"},
    {"vec_host_pin", vec_host_pin, METH_VARARGS, "vec_host_pin(handle, buffer, pages_exclusive): pin a host block that is a mapping of its own (pages_exclusive=1 vouches for that) for in-place access by vec_step_many"},
    {"vec_host_unpin", vec_host_unpin, METH_VARARGS, "vec_host_unpin(handle, buffer)"},
    {"vec_variant", vec_variant, METH_VARARGS, "vec_variant(handle) -> which per-step kernel instantiation and launch choices the env uses, as text (drone_vec_variant)"},
    {"vec_host_transport", vec_host_transport, METH_VARARGS, "vec_host_transport(handle) -> 0 mirror, 1 zero-copy, 2 zero-copy through pinned stand-ins, 3 the same moved by the host copy pool, -1 device buffers"},
    {"vec_sync", vec_sync, METH_VARARGS, "vec_sync(handle): wait until everything enqueued on the env's stream has finished"},
    {"obs_dim", obs_dim, METH_VARARGS, "obs_dim(task) -> floats per observation row"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "drone_binding", "PufferLib-style vec binding of the MI355X drone env (C-ABI: include/drone_vec.h)", -1,
                                       methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit_drone_binding(void) {
    PyObject* m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    PyModule_AddIntConstant(m, "TASK_HOVER", DRONE_TASK_HOVER);
    PyModule_AddIntConstant(m, "TASK_WAYPOINT", DRONE_TASK_WAYPOINT);
    PyModule_AddIntConstant(m, "TASK_SWARM", DRONE_TASK_SWARM);
    PyModule_AddIntConstant(m, "TASK_RACE", DRONE_TASK_RACE);
    PyModule_AddIntConstant(m, "ACT_DIM", DRONE_ACT_DIM);
    return m;
}
