// device_group.hip -- one frame over several GPUs of one node, from one process.
//
// The reference renders on one GPU (reference src/raytrace.cu:509-644 is called once per frame
// from Scene::render, src/scene.cu:209-222).  Its sample loop splits cleanly: every device renders
// a range of the GLOBAL sample indices of every pixel (rt_accumulate_rows; the generator is keyed
// on the global index), the linear (r, g, b, actual_rays) buffers are summed on the first device,
// and the mean + transfer curve run there once (rt_resolve).  More devices than samples per pixel
// also cut the image into bands of rows.
//
//   submit    one host thread per device: a frame is about thirty launches, and queueing them for
//             eight devices from one thread would take longer than a device needs to run its share
//   exchange  RCCL ncclReduce to the first device (single process, one communicator per device,
//             loaded at run time so that the library has no link-time dependency on RCCL), or,
//             where RCCL is unavailable or the same device appears twice, peer copies into staging
//             buffers on the first device followed by additions in device order
//   resolve   on the first device, then one copy to the host
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and enumerators only: the functions are looked up with dlsym

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"

void rt_set_error(const char* fmt, ...);

namespace {

// ---- RCCL, bound at run time ------------------------------------------------

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        auto sym = [&](const char* n) { return dlsym(r.handle, n); };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.CommInitAll && r.CommDestroy && r.Reduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return r;
}

struct Member {
    int device = 0;
    rt_scene* scene = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float* d_sum = nullptr;   // 4 floats per pixel of the whole image
    float* d_stage = nullptr; // on the FIRST device: where this member's sums arrive (peer exchange only)
    ncclComm_t comm = nullptr;
};

#define GROUP_TRY(expr)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RT_ERR_DEVICE;                                                                 \
        }                                                                                         \
    } while (0)

} // namespace

struct rt_device_group {
    std::vector<Member> members;
    bool use_rccl = false;
    size_t pixels = 0;        // capacity of the buffers
    float* d_out = nullptr;   // first device: the resolved image
};

extern "C" void rt_frame_partition(uint32_t rank, uint32_t world, uint32_t spp, uint32_t height, uint32_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (world == 0 || rank >= world) return;
    // sample groups first (perfectly balanced: every device sees all pixels); bands of rows only
    // when there are more devices than samples: groups = the largest divisor of world not above spp
    uint32_t groups = 1;
    for (uint32_t d = 1; d <= world; d++)
        if (world % d == 0 && d <= std::max(spp, 1u)) groups = d;
    const uint32_t bands = world / groups;
    auto share = [](uint32_t r, uint32_t n, uint32_t total, uint32_t& begin, uint32_t& end) {
        const uint32_t base = total / n, extra = total % n;
        begin = r * base + std::min(r, extra);
        end = begin + base + (r < extra ? 1u : 0u);
    };
    share(rank % groups, groups, spp, out[0], out[1]);
    share(rank / groups, bands, height, out[2], out[3]);
}

extern "C" int rt_device_group_free(rt_device_group* g) {
    if (!g) return RT_OK;
    for (Member& m : g->members) {
        (void)hipSetDevice(m.device);
        if (m.comm && rccl().ok) (void)rccl().CommDestroy(m.comm);
        if (m.scene) rt_scene_free(m.scene);
        if (m.d_sum) (void)hipFree(m.d_sum);
        if (m.done) (void)hipEventDestroy(m.done);
        if (m.stream) (void)hipStreamDestroy(m.stream);
    }
    if (!g->members.empty()) {
        (void)hipSetDevice(g->members[0].device);
        for (Member& m : g->members)
            if (m.d_stage) (void)hipFree(m.d_stage);
        if (g->d_out) (void)hipFree(g->d_out);
    }
    delete g;
    return RT_OK;
}

extern "C" int rt_device_group_create(const rt_scene_desc* desc, const int32_t* devices, uint32_t n_devices, rt_device_group** out) {
    if (!out) { rt_set_error("rt_device_group_create: null output"); return RT_ERR_INVALID; }
    *out = nullptr;
    if (!desc || !devices || n_devices == 0 || n_devices > 64) { rt_set_error("rt_device_group_create: bad argument"); return RT_ERR_INVALID; }
    int count = 0, previous = 0;
    GROUP_TRY(hipGetDeviceCount(&count));
    GROUP_TRY(hipGetDevice(&previous));
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= count) { rt_set_error("rt_device_group_create: device %d of %d", devices[i], count); return RT_ERR_INVALID; }
        for (uint32_t j = 0; j < i; j++) distinct = distinct && devices[j] != devices[i];
    }
    rt_device_group* g = new rt_device_group();
    g->members.resize(n_devices);
    int rc = RT_OK;
    for (uint32_t i = 0; i < n_devices && rc == RT_OK; i++) {
        Member& m = g->members[i];
        m.device = devices[i];
        if (hipSetDevice(m.device) != hipSuccess || hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m.done, hipEventDisableTiming) != hipSuccess) {
            rt_set_error("rt_device_group_create: stream or event on device %d: %s", m.device, hipGetErrorString(hipGetLastError()));
            rc = RT_ERR_DEVICE;
            break;
        }
        rc = rt_init(m.device); // also resolves the device's copy of the library's static data
        if (rc == RT_OK) rc = rt_scene_upload(desc, &m.scene);
        if (rc == RT_OK && i > 0 && distinct) { // let the first device read this one's memory and the other way round
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->members[0].device, m.device) == hipSuccess && can) {
                (void)hipSetDevice(g->members[0].device);
                (void)hipDeviceEnablePeerAccess(m.device, 0);
                (void)hipSetDevice(m.device);
                (void)hipDeviceEnablePeerAccess(g->members[0].device, 0);
                (void)hipGetLastError(); // "already enabled" is fine
            }
        }
    }
    // exchange: RCCL where it can be used (more than one device, all distinct, library present)
    const char* want = std::getenv("RT_AMD_EXCHANGE"); // "peer" or "rccl"; default: rccl when possible
    const bool forbid = want && std::strcmp(want, "peer") == 0;
    const bool force = want && std::strcmp(want, "rccl") == 0; // also for a group of one (a one-rank collective: a rehearsal)
    if (rc == RT_OK && (n_devices > 1 || force) && distinct && !forbid && rccl().ok) {
        std::vector<ncclComm_t> comms(n_devices, nullptr);
        std::vector<int> list(devices, devices + n_devices);
        const ncclResult_t r = rccl().CommInitAll(comms.data(), (int)n_devices, list.data());
        if (r == ncclSuccess) {
            for (uint32_t i = 0; i < n_devices; i++) g->members[i].comm = comms[i];
            g->use_rccl = true;
        } else if (force) {
            rt_set_error("rt_device_group_create: ncclCommInitAll: %s", rccl().GetErrorString(r));
            rc = RT_ERR_DEVICE;
        }
    } else if (rc == RT_OK && force) {
        rt_set_error("rt_device_group_create: RCCL exchange requested but %s", !rccl().ok ? "librccl could not be loaded" : "a device appears twice");
        rc = RT_ERR_INVALID;
    }
    (void)hipSetDevice(previous);
    if (rc != RT_OK) {
        const std::string keep = rt_last_error();
        rt_device_group_free(g);
        rt_set_error("%s", keep.c_str());
        return rc;
    }
    *out = g;
    return RT_OK;
}

namespace {

int ensure_buffers(rt_device_group* g, size_t pixels) {
    if (g->pixels >= pixels) return RT_OK;
    const int root = g->members[0].device;
    for (Member& m : g->members) {
        GROUP_TRY(hipSetDevice(m.device));
        if (m.d_sum) (void)hipFree(m.d_sum);
        m.d_sum = nullptr;
        GROUP_TRY(hipMalloc(reinterpret_cast<void**>(&m.d_sum), pixels * 4 * sizeof(float)));
    }
    GROUP_TRY(hipSetDevice(root));
    if (g->d_out) (void)hipFree(g->d_out);
    g->d_out = nullptr;
    GROUP_TRY(hipMalloc(reinterpret_cast<void**>(&g->d_out), pixels * 3 * sizeof(float)));
    if (!g->use_rccl)
        for (size_t i = 1; i < g->members.size(); i++) {
            Member& m = g->members[i];
            if (m.d_stage) (void)hipFree(m.d_stage);
            m.d_stage = nullptr;
            GROUP_TRY(hipMalloc(reinterpret_cast<void**>(&m.d_stage), pixels * 4 * sizeof(float)));
        }
    g->pixels = pixels;
    return RT_OK;
}

} // namespace

extern "C" int rt_device_group_render(rt_device_group* g, const rt_camera* cam, const rt_options* opt, float* out_rgb, rt_stats* stats) {
    if (!g || !cam || !opt || !out_rgb || g->members.empty()) { rt_set_error("rt_device_group_render: null argument"); return RT_ERR_INVALID; }
    if (cam->width == 0 || cam->height == 0 || (uint64_t)cam->width * cam->height > (1ull << 28)) {
        rt_set_error("unsupported resolution %ux%u", cam->width, cam->height);
        return RT_ERR_INVALID;
    }
    int previous = 0;
    GROUP_TRY(hipGetDevice(&previous));
    const size_t pixels = (size_t)cam->width * cam->height;
    int rc = ensure_buffers(g, pixels);
    if (rc != RT_OK) { (void)hipSetDevice(previous); return rc; }
    const uint32_t n = (uint32_t)g->members.size();
    const size_t sum_bytes = pixels * 4 * sizeof(float);

    // submit: every device renders its share into its own zeroed buffer
    std::vector<int> codes(n, RT_OK);
    std::vector<std::string> messages(n);
    auto submit = [&](uint32_t i) {
        Member& m = g->members[i];
        uint32_t part[4];
        rt_frame_partition(i, n, cam->num_rays, cam->height, part);
        int c = RT_OK;
        if (hipSetDevice(m.device) != hipSuccess || hipMemsetAsync(m.d_sum, 0, sum_bytes, m.stream) != hipSuccess) {
            c = RT_ERR_DEVICE;
            messages[i] = std::string("device ") + std::to_string(m.device) + ": " + hipGetErrorString(hipGetLastError());
        } else {
            c = rt_accumulate_rows(m.scene, cam, opt, part[0], part[1], part[2], part[3], m.d_sum, m.stream, i == 0 ? stats : nullptr);
            if (c != RT_OK) messages[i] = rt_last_error(); // thread-local: fetch it on this thread
            else if (hipEventRecord(m.done, m.stream) != hipSuccess) { c = RT_ERR_DEVICE; messages[i] = "hipEventRecord failed"; }
        }
        codes[i] = c;
    };
    if (n == 1) {
        submit(0);
    } else {
        std::vector<std::thread> threads;
        for (uint32_t i = 0; i < n; i++) threads.emplace_back(submit, i);
        for (std::thread& t : threads) t.join();
    }
    for (uint32_t i = 0; i < n; i++)
        if (codes[i] != RT_OK) {
            for (Member& m : g->members) { (void)hipSetDevice(m.device); (void)hipStreamSynchronize(m.stream); }
            (void)hipSetDevice(previous);
            rt_set_error("%s", messages[i].c_str());
            return codes[i];
        }

    // exchange: the first device's buffer receives the sum over the devices
    Member& root = g->members[0];
    if (g->use_rccl) {
        const Rccl& r = rccl();
        ncclResult_t res = r.GroupStart();
        for (uint32_t i = 0; i < n && res == ncclSuccess; i++) {
            Member& m = g->members[i];
            (void)hipSetDevice(m.device);
            res = r.Reduce(m.d_sum, m.d_sum, pixels * 4, ncclFloat, ncclSum, 0, m.comm, m.stream); // in place on the root
        }
        const ncclResult_t end = r.GroupEnd();
        if (res == ncclSuccess) res = end;
        if (res != ncclSuccess) {
            // nothing of this frame may still run on a member's stream when the caller reuses or frees the buffers
            for (Member& m : g->members) { (void)hipSetDevice(m.device); (void)hipStreamSynchronize(m.stream); }
            (void)hipSetDevice(previous);
            rt_set_error("rt_device_group_render: RCCL reduce: %s", r.GetErrorString(res));
            return RT_ERR_DEVICE;
        }
    } else if (n > 1) {
        GROUP_TRY(hipSetDevice(root.device));
        for (uint32_t i = 1; i < n; i++) { // additions in device order: the image does not depend on timing
            Member& m = g->members[i];
            GROUP_TRY(hipStreamWaitEvent(root.stream, m.done, 0));
            GROUP_TRY(hipMemcpyPeerAsync(m.d_stage, root.device, m.d_sum, m.device, sum_bytes, root.stream));
            if ((rc = rt_add_sums(root.d_sum, m.d_stage, (uint32_t)(pixels * 4), root.stream)) != RT_OK) { (void)hipSetDevice(previous); return rc; }
        }
    }
    GROUP_TRY(hipSetDevice(root.device));
    if ((rc = rt_resolve(root.d_sum, (uint32_t)pixels, g->d_out, root.stream)) != RT_OK) { (void)hipSetDevice(previous); return rc; }
    GROUP_TRY(hipMemcpyAsync(out_rgb, g->d_out, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, root.stream));
    GROUP_TRY(hipStreamSynchronize(root.stream));
    if (g->use_rccl) // the other devices' part of the collective
        for (uint32_t i = 1; i < n; i++) {
            GROUP_TRY(hipSetDevice(g->members[i].device));
            GROUP_TRY(hipStreamSynchronize(g->members[i].stream));
        }
    GROUP_TRY(hipSetDevice(previous));
    return RT_OK;
}
