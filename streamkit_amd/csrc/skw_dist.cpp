// skw_dist.cpp — include/skw_dist.h: the Oneshot path's transcript gather as a C ABI over RCCL.  Host C++ (hipcc only for the HIP runtime headers): no kernels.
#include "../../include/skw_dist.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

static_assert(SKW_DIST_ID_BYTES == sizeof(ncclUniqueId), "the id handed between ranks is RCCL's ncclUniqueId");

struct skw_dist {
    int world = 0, rank0 = 0;                       // ranks in the group; the group rank of local rank 0 (local rank i is group rank rank0 + i)
    std::vector<int> dev; std::vector<ncclComm_t> comm; std::vector<hipStream_t> stream;
    std::vector<int32_t*> d_send, d_recv; size_t cap_rows = 0;      // device staging, grown on demand
    char errbuf[512] = {0};
};
static void set_err(char* e, size_t n, const char* fmt, ...) { if (!e || !n) return; va_list ap; va_start(ap, fmt); vsnprintf(e, n, fmt, ap); va_end(ap); }
#define DIST_HIP(x, where, ret) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(where, 512, "%s: %s", #x, hipGetErrorString(e_)); return ret; } } while (0)
#define DIST_NCCL(x, where, ret) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { set_err(where, 512, "%s: %s", #x, ncclGetErrorString(r_)); return ret; } } while (0)

static bool add_streams(skw_dist* d, char* err) {
    for (size_t i = 0; i < d->dev.size(); ++i) {
        DIST_HIP(hipSetDevice(d->dev[i]), err, false);
        hipStream_t s = nullptr; DIST_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), err, false);
        d->stream.push_back(s); d->d_send.push_back(nullptr); d->d_recv.push_back(nullptr);
    }
    return true;
}
extern "C" skw_dist* skw_dist_create_local(const int* devices, int n, char* err, size_t errlen) {
    char e[512] = {0};
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { set_err(err, errlen, "no HIP device available: the transcript gather runs over RCCL between MI355X GPUs"); return nullptr; }
    if (!devices || n < 1 || n > n_dev) { set_err(err, errlen, "skw_dist_create_local: %d devices asked for, %d visible", n, n_dev); return nullptr; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j)
            if ((j < i && devices[i] == devices[j]) || devices[i] < 0 || devices[i] >= n_dev) {
                set_err(err, errlen, "skw_dist_create_local: device list must name %d different visible devices", n); return nullptr; }
    skw_dist* d = new skw_dist(); d->world = n; d->rank0 = 0; d->dev.assign(devices, devices + n); d->comm.assign(n, nullptr);
    ncclResult_t r = ncclCommInitAll(d->comm.data(), n, d->dev.data());
    if (r != ncclSuccess) { set_err(err, errlen, "ncclCommInitAll: %s", ncclGetErrorString(r)); d->comm.clear(); skw_dist_free(d); return nullptr; }
    if (!add_streams(d, e)) { set_err(err, errlen, "%s", e); skw_dist_free(d); return nullptr; }
    return d;
}
extern "C" int skw_dist_unique_id(unsigned char id[SKW_DIST_ID_BYTES], char* err, size_t errlen) {
    ncclUniqueId u; ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) { set_err(err, errlen, "ncclGetUniqueId: %s", ncclGetErrorString(r)); return -1; }
    memcpy(id, &u, sizeof u); return 0;
}
extern "C" skw_dist* skw_dist_create_rank(const unsigned char id[SKW_DIST_ID_BYTES], int rank, int world, int device, char* err, size_t errlen) {
    char e[512] = {0};
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { set_err(err, errlen, "no HIP device available: the transcript gather runs over RCCL between MI355X GPUs"); return nullptr; }
    if (!id || world < 1 || rank < 0 || rank >= world || device < 0 || device >= n_dev) {
        set_err(err, errlen, "skw_dist_create_rank: rank %d of %d on device %d (%d visible)", rank, world, device, n_dev); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_err(err, errlen, "hipSetDevice(%d) failed", device); return nullptr; }
    skw_dist* d = new skw_dist(); d->world = world; d->rank0 = rank; d->dev.assign(1, device); d->comm.assign(1, nullptr);
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    ncclResult_t r = ncclCommInitRank(&d->comm[0], world, u, rank);
    if (r != ncclSuccess) { set_err(err, errlen, "ncclCommInitRank: %s", ncclGetErrorString(r)); d->comm.clear(); skw_dist_free(d); return nullptr; }
    if (!add_streams(d, e)) { set_err(err, errlen, "%s", e); skw_dist_free(d); return nullptr; }
    return d;
}
extern "C" int skw_dist_world(const skw_dist* d) { return d ? d->world : 0; }
extern "C" int skw_dist_n_local(const skw_dist* d) { return d ? (int)d->dev.size() : 0; }
extern "C" const char* skw_dist_last_error(const skw_dist* d) { return d ? d->errbuf : "null handle"; }
extern "C" int skw_dist_all_gather_tokens(skw_dist* d, const int32_t* const* send, int rows, int32_t* const* recv) {
    if (!d) return -1;
    char* err = d->errbuf; err[0] = 0;
    const int nl = (int)d->dev.size();
    if (!send || !recv || rows < 1) { set_err(err, 512, "skw_dist_all_gather_tokens: bad arguments"); return -1; }
    for (int i = 0; i < nl; ++i) if (!send[i] || !recv[i]) { set_err(err, 512, "skw_dist_all_gather_tokens: null buffer for local rank %d", i); return -1; }
    const size_t count = (size_t)rows * SKW_DIST_ROW;
    if ((size_t)rows > d->cap_rows) {
        for (int i = 0; i < nl; ++i) {
            DIST_HIP(hipSetDevice(d->dev[i]), err, -1);
            (void)hipFree(d->d_send[i]); (void)hipFree(d->d_recv[i]); d->d_send[i] = d->d_recv[i] = nullptr;
            DIST_HIP(hipMalloc((void**)&d->d_send[i], count * 4), err, -1);
            DIST_HIP(hipMalloc((void**)&d->d_recv[i], count * 4 * d->world), err, -1);
        }
        d->cap_rows = rows;
    }
    for (int i = 0; i < nl; ++i) { DIST_HIP(hipSetDevice(d->dev[i]), err, -1); DIST_HIP(hipMemcpyAsync(d->d_send[i], send[i], count * 4, hipMemcpyHostToDevice, d->stream[i]), err, -1); }
    DIST_NCCL(ncclGroupStart(), err, -1);      // one group: the n local ranks' calls are issued together (a single-threaded host would deadlock on the first otherwise)
    for (int i = 0; i < nl; ++i) {
        ncclResult_t r = ncclAllGather(d->d_send[i], d->d_recv[i], count, ncclInt32, d->comm[i], d->stream[i]);
        if (r != ncclSuccess) { (void)ncclGroupEnd(); set_err(err, 512, "ncclAllGather (local rank %d): %s", i, ncclGetErrorString(r)); return -1; }
    }
    DIST_NCCL(ncclGroupEnd(), err, -1);
    for (int i = 0; i < nl; ++i) { DIST_HIP(hipSetDevice(d->dev[i]), err, -1); DIST_HIP(hipMemcpyAsync(recv[i], d->d_recv[i], count * 4 * d->world, hipMemcpyDeviceToHost, d->stream[i]), err, -1); }
    for (int i = 0; i < nl; ++i) { DIST_HIP(hipSetDevice(d->dev[i]), err, -1); DIST_HIP(hipStreamSynchronize(d->stream[i]), err, -1); }
    return 0;
}
extern "C" void skw_dist_free(skw_dist* d) {
    if (!d) return;
    for (size_t i = 0; i < d->dev.size(); ++i) {
        (void)hipSetDevice(d->dev[i]);
        if (i < d->stream.size() && d->stream[i]) { (void)hipStreamSynchronize(d->stream[i]); (void)hipStreamDestroy(d->stream[i]); }
        if (i < d->d_send.size()) { (void)hipFree(d->d_send[i]); (void)hipFree(d->d_recv[i]); }
        if (i < d->comm.size() && d->comm[i]) (void)ncclCommDestroy(d->comm[i]);
    }
    delete d;
}
