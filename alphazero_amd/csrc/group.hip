// group.hip -- one process, every GPU of the node: the reference's single generate process
// (generate.cpp:37-123) scaled out by sharding the games over GPUs.  Trees never interact, so the data
// path has no collective; RCCL carries exactly the weight hand-off (load_network, util.h:60-105, read once
// by the process and broadcast to every GPU's copy of the blob).  Host code only.
//
//   * one engine context (api.hip) per (GPU, context slot); game_slot_base numbers the games of the group
//   * one host thread per GPU issues that GPU's launches (a context is driven by one thread at a time)
//   * ncclCommInitAll over the group's GPUs; az_group_load_weights = one H2D copy on the first GPU,
//     ncclBroadcast of the raw fp32 blob, BN fold + pack on every GPU (weights.hip) -- no host bounce
//   * finished replay rows are drained per context and merged on the host in context order
// RCCL is bound at run time (dlopen of librccl.so.1, the copy already mapped if there is one; AZ_RCCL_LIB names
// another file), and only by a group of more than one GPU: the library has no link-time dependency on it, and a
// one-GPU group -- what the generator is on a single-GPU box -- neither loads it nor creates a communicator: its
// blob goes host -> device and is folded there.
#include "../../include/az_hip.h"
#include "az_worker.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" int az_internal_fail(int code, const char* msg);      // api.hip: records the message for az_last_error

namespace {

// the six RCCL entry points this path uses (rccl.h), bound by name
typedef void* nccl_comm_t;
struct Rccl {
    void* so = nullptr;
    int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int /*ncclDataType_t*/, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
    bool load() {
        if (so) return true;
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        const char* forced = getenv("AZ_RCCL_LIB");
        if (forced) so = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        else for (const char* n : names) { so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
        if (!so) { const char* de = dlerror(); why = std::string("librccl not found: ") + (de ? de : "?"); return false; }
#define BIND(field, sym) do { *(void**)(&field) = dlsym(so, sym); if (!field) { why = std::string("missing RCCL symbol ") + sym; return false; } } while (0)
        BIND(CommInitAll, "ncclCommInitAll"); BIND(CommDestroy, "ncclCommDestroy"); BIND(GroupStart, "ncclGroupStart");
        BIND(GroupEnd, "ncclGroupEnd"); BIND(Broadcast, "ncclBroadcast"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
enum { NCCL_FLOAT32 = 7 };                                        // ncclFloat32 (rccl.h)

}  // namespace

struct az_group {
    std::vector<int> devices;
    int per_device = 1;
    std::vector<az_ctx*> ctxs;                  // [device][slot], device-major
    std::vector<nccl_comm_t> comms;             // one per device
    std::vector<float*> blob_dev;               // one fp32 blob per device: the broadcast's buffer
    std::vector<az_worker::Worker*> workers;               // one per device
    size_t n_floats = 0;
};

static int gfail(int code, const std::string& m) { return az_internal_fail(code, m.c_str()); }
#define GHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return gfail(AZ_E_HIP, std::string(#call " failed: ") + hipGetErrorString(e_)); } while (0)
#define GNCCL(call) do { int r_ = (call); if (r_ != 0) \
    return gfail(AZ_E_HIP, std::string(#call " failed: ") + g_rccl.GetErrorString(r_)); } while (0)

// run job(device index) on every GPU's thread, return the first failure
static int on_every_device(az_group* g, const std::function<int(int)>& job) {
    for (size_t d = 0; d < g->devices.size(); ++d) g->workers[d]->post([job, d] { return job((int)d); });
    int rc = AZ_OK;
    std::string first;
    for (size_t d = 0; d < g->devices.size(); ++d) {
        std::string m;
        const int r = g->workers[d]->wait(m);
        if (r < 0 && rc == AZ_OK) { rc = r; first = "GPU " + std::to_string(g->devices[d]) + ": " + m; }
    }
    return rc == AZ_OK ? AZ_OK : gfail(rc, first);
}

extern "C" int az_group_destroy(az_group* g) {
    if (!g) return AZ_OK;
    for (az_worker::Worker* w : g->workers) delete w;                       // joins: no launch is in flight on the host side
    for (az_ctx* c : g->ctxs) az_ctx_destroy(c);
    for (size_t d = 0; d < g->comms.size(); ++d) if (g->comms[d]) g_rccl.CommDestroy(g->comms[d]);
    for (size_t d = 0; d < g->blob_dev.size(); ++d)
        if (g->blob_dev[d]) { (void)hipSetDevice(g->devices[d]); (void)hipFree(g->blob_dev[d]); }
    delete g;
    return AZ_OK;
}

extern "C" int az_group_create(az_group** out, const az_config* cfg, const int* devices, int n_devices, int contexts_per_device) {
    if (!out || !cfg || !devices) return gfail(AZ_E_ARG, "null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return gfail(AZ_E_ARG, "n_devices " + std::to_string(n_devices) + " out of range (1..64)");
    if (contexts_per_device < 1 || contexts_per_device > 8) return gfail(AZ_E_ARG, "contexts_per_device out of range (1..8)");
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return gfail(AZ_E_ARG, "device " + std::to_string(devices[i]) + " listed twice");
    if (n_devices > 1) {                                         // a one-GPU group has nothing to broadcast to
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!g_rccl.load()) return gfail(AZ_E_HIP, g_rccl.why);
    }
    az_group* g = new az_group;
    g->devices.assign(devices, devices + n_devices);
    g->per_device = contexts_per_device;
    g->comms.assign((size_t)n_devices, nullptr);
    g->blob_dev.assign((size_t)n_devices, nullptr);
    int rc = AZ_OK;
    for (int d = 0; d < n_devices && rc == AZ_OK; ++d)
        for (int s = 0; s < contexts_per_device && rc == AZ_OK; ++s) {
            az_config c = *cfg;
            c.device = devices[d];
            c.game_slot_base = cfg->game_slot_base + (d * contexts_per_device + s) * cfg->G;
            az_ctx* ctx = nullptr;
            rc = az_ctx_create(&ctx, &c);
            if (rc == AZ_OK) g->ctxs.push_back(ctx);
        }
    if (rc != AZ_OK) { const std::string keep = az_last_error(); az_group_destroy(g); return gfail(rc, keep); }
    g->n_floats = az_weights_size(g->ctxs[0]);
    for (int d = 0; d < n_devices; ++d) {
        hipError_t e = hipSetDevice(devices[d]);
        if (e == hipSuccess && g->n_floats) e = hipMalloc((void**)&g->blob_dev[(size_t)d], g->n_floats * sizeof(float));
        if (e != hipSuccess) { az_group_destroy(g); return gfail(AZ_E_HIP, std::string("blob buffer: ") + hipGetErrorString(e)); }
    }
    if (n_devices > 1) {
        const int nr = g_rccl.CommInitAll(g->comms.data(), n_devices, devices);
        if (nr != 0) { const std::string m = g_rccl.GetErrorString(nr); az_group_destroy(g); return gfail(AZ_E_HIP, "ncclCommInitAll failed: " + m); }
    }
    for (int d = 0; d < n_devices; ++d) g->workers.push_back(new az_worker::Worker(az_last_error));
    *out = g;
    return AZ_OK;
}

extern "C" int az_group_contexts(const az_group* g) { return g ? (int)g->ctxs.size() : 0; }
extern "C" az_ctx* az_group_context(az_group* g, int i) {
    return g && i >= 0 && i < (int)g->ctxs.size() ? g->ctxs[(size_t)i] : nullptr;
}

extern "C" int az_group_load_weights(az_group* g, const float* blob, size_t n_floats) {
    if (!g || !blob) return gfail(AZ_E_ARG, "null argument");
    if (n_floats != g->n_floats || n_floats == 0) return gfail(AZ_E_ARG, "blob has " + std::to_string(n_floats) + " floats, expected " + std::to_string(g->n_floats));
    const int nd = (int)g->devices.size();
    // the blob enters the node once, on the first GPU, on that GPU's first context's stream ...
    GHIP(hipSetDevice(g->devices[0]));
    hipStream_t s0 = (hipStream_t)az_stream(g->ctxs[0]);
    GHIP(hipMemcpyAsync(g->blob_dev[0], blob, n_floats * sizeof(float), hipMemcpyHostToDevice, s0));
    // ... and reaches every other GPU over xGMI: one broadcast, each GPU on its own first context's stream
    if (nd > 1) {
        GNCCL(g_rccl.GroupStart());
        for (int d = 0; d < nd; ++d) {
            hipStream_t s = (hipStream_t)az_stream(g->ctxs[(size_t)d * g->per_device]);
            const int r = g_rccl.Broadcast(g->blob_dev[(size_t)d], g->blob_dev[(size_t)d], n_floats, NCCL_FLOAT32, 0, g->comms[(size_t)d], s);
            if (r != 0) { g_rccl.GroupEnd(); return gfail(AZ_E_HIP, std::string("ncclBroadcast failed: ") + g_rccl.GetErrorString(r)); }
        }
        GNCCL(g_rccl.GroupEnd());
    }
    // every GPU folds and packs its own copy (weights.hip), all contexts of the GPU from the same blob
    return on_every_device(g, [g, n_floats](int d) {
        hipError_t e = hipSetDevice(g->devices[(size_t)d]);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)az_stream(g->ctxs[(size_t)d * g->per_device]));
        if (e != hipSuccess) return gfail(AZ_E_HIP, std::string("broadcast wait: ") + hipGetErrorString(e));
        for (int s = 0; s < g->per_device; ++s) {
            const int r = az_load_weights_dev(g->ctxs[(size_t)d * g->per_device + s], g->blob_dev[(size_t)d], n_floats);
            if (r != AZ_OK) return r;
        }
        return (int)AZ_OK;
    });
}

extern "C" int az_group_selfplay_begin(az_group* g, int iter_budget) {
    if (!g) return gfail(AZ_E_ARG, "null argument");
    return on_every_device(g, [g, iter_budget](int d) {
        for (int s = 0; s < g->per_device; ++s) {
            const int r = az_selfplay_begin(g->ctxs[(size_t)d * g->per_device + s], iter_budget);
            if (r != AZ_OK) return r;
        }
        return (int)AZ_OK;
    });
}

// Launches are asynchronous: each GPU's thread feeds its contexts in slices so that every stream stays busy.
extern "C" int az_group_selfplay_run(az_group* g, int rounds) {
    if (!g) return gfail(AZ_E_ARG, "null argument");
    return on_every_device(g, [g, rounds](int d) {
        for (int done = 0; done < rounds;) {
            const int k = rounds - done < 16 ? rounds - done : 16;
            for (int s = 0; s < g->per_device; ++s) {
                const int r = az_selfplay_run(g->ctxs[(size_t)d * g->per_device + s], k);
                if (r != AZ_OK) return r;
            }
            done += k;
        }
        return (int)AZ_OK;
    });
}

extern "C" int az_group_sync(az_group* g) {
    if (!g) return gfail(AZ_E_ARG, "null argument");
    return on_every_device(g, [g](int d) {
        for (int s = 0; s < g->per_device; ++s) {
            const int r = az_sync(g->ctxs[(size_t)d * g->per_device + s]);
            if (r != AZ_OK) return r;
        }
        return (int)AZ_OK;
    });
}

// The merge of per-source replay rows into one caller buffer: sources are drained in order, each into the
// rows behind the previous source's, until `cap` rows are out; what does not fit stays in its source for the
// next call; a source's error ends the merge and is returned.  Pure host code (the sources are callbacks), so
// the ordering logic is testable without a GPU.
extern "C" int az_merge_drain(int n_src, az_drain_fn fn, void* const* srcs, int A, int cap,
                              uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    if (n_src < 0 || !fn || (n_src > 0 && !srcs) || A < 1) return gfail(AZ_E_ARG, "bad argument to az_merge_drain");
    if (cap < 0) return gfail(AZ_E_ARG, "cap is negative");
    if (cap > 0 && (!cells || !to_move || !policy || !reward)) return gfail(AZ_E_ARG, "null row buffer");
    int total = 0;
    for (int i = 0; i < n_src && total < cap; ++i) {
        const int n = fn(srcs[i], cap - total, cells + (size_t)total * A, to_move + total,
                         policy + (size_t)total * A, reward + (size_t)total * 2);
        if (n < 0) return n;
        if (n > cap - total) return gfail(AZ_E_STATE, "source " + std::to_string(i) + " returned more rows than asked for");
        total += n;
    }
    return total;
}

static int drain_ctx(void* src, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    return az_replay_drain((az_ctx*)src, cap, cells, to_move, policy, reward);
}

// Rows of every context, in context order (GPU-major): the gather of replay positions to the one process
// that owns the PUSH socket (replay.hpp:41-54).  At ~1.6 KB per position the rate is MB/s: plain per-GPU
// D2H copies, merged on the host.
extern "C" int az_group_replay_drain(az_group* g, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    if (!g) return gfail(AZ_E_ARG, "null argument");
    return az_merge_drain((int)g->ctxs.size(), drain_ctx, (void* const*)g->ctxs.data(), az_ctx_cells(g->ctxs[0]), cap,
                          cells, to_move, policy, reward);
}

static int drain_retired_ctx(void* src, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    return az_replay_drain_retired((az_ctx*)src, cap, cells, to_move, policy, reward);
}

// The two halves of the overlapped hand-over (az_replay_flip / az_replay_drain_retired) for every context: the flip
// only records an event and swaps pointers, so the calling thread does it; the drain merges in context order.
extern "C" int az_group_replay_flip(az_group* g) {
    if (!g) return gfail(AZ_E_ARG, "null argument");
    for (az_ctx* c : g->ctxs) { const int r = az_replay_flip(c); if (r != AZ_OK) return r; }
    return AZ_OK;
}
extern "C" int az_group_replay_drain_retired(az_group* g, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    if (!g) return gfail(AZ_E_ARG, "null argument");
    return az_merge_drain((int)g->ctxs.size(), drain_retired_ctx, (void* const*)g->ctxs.data(), az_ctx_cells(g->ctxs[0]), cap,
                          cells, to_move, policy, reward);
}

extern "C" int az_group_get_stats(az_group* g, az_stats* out) {
    if (!g || !out) return gfail(AZ_E_ARG, "null argument");
    memset(out, 0, sizeof *out);
    for (az_ctx* c : g->ctxs) {
        az_stats s;
        const int r = az_get_stats(c, &s);
        if (r != AZ_OK) return r;
        out->sims += s.sims; out->evals += s.evals; out->moves += s.moves; out->games += s.games;
        out->replay_rows += s.replay_rows; out->replay_dropped += s.replay_dropped;
        out->error_games += s.error_games; out->tree_restarts += s.tree_restarts; out->sims_backed_up += s.sims_backed_up;
        out->descents += s.descents; out->tree_levels += s.tree_levels;
        out->children_scanned += s.children_scanned; out->children_expanded += s.children_expanded;
        out->saturated += s.saturated;
        out->rounds = s.rounds > out->rounds ? s.rounds : out->rounds;
        out->max_nodes = s.max_nodes > out->max_nodes ? s.max_nodes : out->max_nodes;
        out->max_edges = s.max_edges > out->max_edges ? s.max_edges : out->max_edges;
    }
    return AZ_OK;
}
