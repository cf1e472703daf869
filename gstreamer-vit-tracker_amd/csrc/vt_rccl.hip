// vt_rccl.hip — the path's one collective: the start-up weight broadcast, callable from a host without Python.
#include "vt_engine.hpp"
#include <dlfcn.h>

// ---- RCCL start-up broadcast (librccl loaded lazily) -----------------------------------------------

namespace {
struct NcclId { char internal[VT_RCCL_ID_BYTES]; };   // ≙ ncclUniqueId
typedef void* NcclComm;
struct RcclApi {
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi load_rccl() {
    RcclApi api;
    void* h = RTLD_DEFAULT;                          // a copy already loaded by the host wins
    if (!dlsym(h, "ncclGetUniqueId")) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        h = nullptr;
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return api;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.Broadcast = (decltype(api.Broadcast))dlsym(h, "ncclBroadcast");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.Broadcast && api.CommDestroy;
    return api;
}
// one host thread per GPU may call in at the same time: a C++11 magic static hands every caller the
// fully built table (initialisation runs once, the others wait for it)
RcclApi* rccl_api() {
    static RcclApi api = load_rccl();
    return api.ok ? &api : nullptr;
}
int rccl_err(RcclApi* a, const char* what, int code) {
    return set_err(VT_ERR_HIP, "%s failed: %s (ncclResult %d)", what,
                   a->GetErrorString ? a->GetErrorString(code) : "?", code);
}
}  // namespace

extern "C" {

int vt_rccl_unique_id(uint8_t id_out[VT_RCCL_ID_BYTES]) try {
    if (!id_out) return set_err(VT_ERR_INVALID_ARG, "null id buffer");
    RcclApi* a = rccl_api();
    if (!a) return set_err(VT_ERR_NO_DEVICE, "librccl could not be loaded (dlopen librccl.so.1 / librccl.so)");
    NcclId id;
    memset(&id, 0, sizeof(id));
    if (int rc = a->GetUniqueId(&id)) return rccl_err(a, "ncclGetUniqueId", rc);
    memcpy(id_out, &id, sizeof(id));
    return VT_OK;
} VT_NOTHROW_INT

int vt_broadcast_weights_rccl(const uint8_t id[VT_RCCL_ID_BYTES], int world, int rank, int device_id,
                              const char* weights_path, void** d_blob_out, size_t* bytes_out) try {
    if (!id || !d_blob_out || !bytes_out || world < 1 || rank < 0 || rank >= world)
        return set_err(VT_ERR_INVALID_ARG, "bad argument");
    *d_blob_out = nullptr; *bytes_out = 0;
    if (rank == 0 && !weights_path) return set_err(VT_ERR_INVALID_ARG, "rank 0 needs the weights path");
    if (int rc = check_device(device_id)) return rc;
    RcclApi* a = rccl_api();
    if (!a) return set_err(VT_ERR_NO_DEVICE, "librccl could not be loaded");
    DEVICE_SCOPE(device_id);
    std::vector<uint8_t> blob;
    if (rank == 0)
        if (int rc = read_file(weights_path, &blob)) return rc;
    NcclId nid;
    memcpy(&nid, id, sizeof(nid));
    NcclComm comm = nullptr;
    if (int rc = a->CommInitRank(&comm, world, nid, rank)) return rccl_err(a, "ncclCommInitRank", rc);
    hipStream_t st = nullptr;
    unsigned long long* d_n = nullptr;
    void* d_blob = nullptr;
    int ret = VT_OK;
    auto fail = [&](int code) { ret = code; };
    do {
        hipError_t he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (he != hipSuccess) { fail(set_err(VT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(he))); break; }
        if ((he = hipMalloc((void**)&d_n, 8)) != hipSuccess) { fail(set_err(VT_ERR_OOM, "hipMalloc: %s", hipGetErrorString(he))); break; }
        unsigned long long n = blob.size();
        if ((he = hipMemcpyAsync(d_n, &n, 8, hipMemcpyHostToDevice, st)) != hipSuccess) { fail(set_err(VT_ERR_HIP, "copy: %s", hipGetErrorString(he))); break; }
        if (int rc = a->Broadcast(d_n, d_n, 8, /*ncclUint8*/ 1, 0, comm, st)) { fail(rccl_err(a, "ncclBroadcast(size)", rc)); break; }
        if ((he = hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
            (he = hipStreamSynchronize(st)) != hipSuccess) { fail(set_err(VT_ERR_HIP, "size read-back: %s", hipGetErrorString(he))); break; }
        if (n < kHeaderBytes || n > (1ull << 36)) { fail(set_err(VT_ERR_FORMAT, "broadcast blob size %llu out of range", n)); break; }
        if ((he = hipMalloc(&d_blob, n)) != hipSuccess) { fail(set_err(VT_ERR_OOM, "hipMalloc(%llu): %s", n, hipGetErrorString(he))); break; }
        if (rank == 0 && (he = hipMemcpyAsync(d_blob, blob.data(), n, hipMemcpyHostToDevice, st)) != hipSuccess) {
            fail(set_err(VT_ERR_HIP, "blob upload: %s", hipGetErrorString(he))); break; }
        // one message: a single large transfer suits xGMI's per-link bandwidth better than many small ones
        if (int rc = a->Broadcast(d_blob, d_blob, n, 1, 0, comm, st)) { fail(rccl_err(a, "ncclBroadcast(blob)", rc)); break; }
        if ((he = hipStreamSynchronize(st)) != hipSuccess) { fail(set_err(VT_ERR_HIP, "broadcast: %s", hipGetErrorString(he))); break; }
        *d_blob_out = d_blob; *bytes_out = (size_t)n;
        d_blob = nullptr;
    } while (0);
    if (d_blob) (void)hipFree(d_blob);
    if (d_n) (void)hipFree(d_n);
    if (st) (void)hipStreamDestroy(st);
    (void)a->CommDestroy(comm);
    return ret;
} VT_NOTHROW_INT

void vt_free_device_blob(int device_id, void* d_blob) try {
    if (!d_blob) return;
    DeviceScope ds(device_id);
    (void)hipFree(d_blob);
} VT_NOTHROW_VOID

}  // extern "C"
