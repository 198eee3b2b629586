// Step plans: the library's own executor for a captured step (include/vhap_hip.h, "Step plans").
//
// A fit step is ~45 kernel launches on up to three concurrent chains, identical for the 50-500 steps of a stage
// (vhap/model/tracker.py:1391-1416 calls optimize_iter that often).  The host records the step ONCE under HIP stream capture (the
// Python orchestration in vhap_amd/step.py stays what it is) -- and instead of instantiating the captured graph and handing it to
// hipGraphLaunch, this file walks the graph (kernel nodes + dependency edges), lays the nodes out over a fixed set of streams of the
// plan's own and replays them with plain hipLaunchKernel calls, cross-stream edges as event record / wait pairs:
//
//   * the stream layout is OURS and fixed at build time: a node's first-captured successor stays on the node's stream (the main chain
//     stays on the launch stream), every other successor goes to a side stream -- hipGraphLaunch re-partitions the DAG on every
//     instantiate and, on ROCm 7, dereferences garbage when the launch stream shares a hardware queue with two of its internal
//     branch streams (profiles/r02_graph_launch_crash.txt), which is why round 2's test suite ran a different configuration than it shipped;
//   * one C call per step: the host cost is ~45 hipLaunchKernel + ~25 event operations from one tight native loop;
//   * any node can be bracketed by timing events (vhap_plan_launch_timed): the in-step duration of a kernel without a profiler and
//     without in-kernel clock stamps (HIP refuses to read event-record nodes of a hipGraph replay).
//
// Only what a captured step contains is supported: kernel, memset (1-D / 2-D), flat device-to-device memcpy and empty nodes; anything else ->
// VHAP_E_UNSUPPORTED and the caller keeps the graph.  The plan borrows the graph's kernel-argument storage: the hipGraph_t must outlive the plan.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_ext.h>

#include "common.h"

namespace {

struct PlanNode {
    int type = 0;                       // 0 kernel, 1 memset, 2 empty, 3 memcpy (1-D, device to device)
    int raw = 0;                        // index in the graph's own node list
    hipGraphNode_t handle = nullptr;    // the captured graph's node (identity only: what a host that watched the capture grow knows it by)
    hipKernelNodeParams kp{};
    hipMemsetParams ms{};
    void* cp_dst = nullptr;
    const void* cp_src = nullptr;
    size_t cp_bytes = 0;
    int stream = 0;                     // 0 = the launch stream, k > 0 = plan stream k - 1
    std::vector<int> waits;             // events to wait for before the launch
    int record = -1;                    // event to record behind the launch
    std::vector<int> deps;              // predecessor nodes (launch order indices)
    bool open_tail = false;             // side-stream node with no path to a later node of stream 0 (see vhap_plan_launch, DEFER_JOIN)
    std::string name;
};

}  // namespace

struct vhap_plan {
    std::vector<PlanNode> nodes;        // launch order (a topological order of the DAG)
    std::vector<hipStream_t> streams;   // plan-owned side streams
    std::vector<hipEvent_t> events;     // cross-stream edges
    hipEvent_t start = nullptr;         // recorded on the launch stream at the head of a replay: the side streams' roots wait for it
    std::vector<hipEvent_t> tails;      // one per side stream: the launch stream waits for them at the end of a replay
    std::vector<hipEvent_t> tev;        // timing events (2 per node + 1), created by the first timed launch
    bool tails_open = false;            // the last launch deferred its join
    bool tails_recorded = true;         // ... and recorded its tail events (false: vhap_plan_join records them)
    int device = 0;
};

namespace {

const char* node_type_name(hipGraphNodeType t) {
    switch (t) {
        case hipGraphNodeTypeKernel: return "kernel";
        case hipGraphNodeTypeMemcpy: return "memcpy";
        case hipGraphNodeTypeMemset: return "memset";
        case hipGraphNodeTypeHost: return "host";
        case hipGraphNodeTypeGraph: return "child graph";
        case hipGraphNodeTypeEmpty: return "empty";
        case hipGraphNodeTypeWaitEvent: return "event wait";
        case hipGraphNodeTypeEventRecord: return "event record";
        default: return "other";
    }
}

// The plans' side streams come from ONE pool per host thread and device (side stream k of every plan of this thread is the same HIP
// stream), created on first use and kept for the life of the thread.  HIP multiplexes a process's streams onto four hardware queues and two
// streams on one queue run in series: with streams of its own per plan, the four plans of a sharded step (forward / pixel + texture /
// geometry / Adam) held eight, the texture gradient's side chain landed on the LAUNCH stream's queue and the reduce-scatter queued up in
// front of the geometry plan it was meant to run under (profiles/r05_call9_sharded_step_timeline.txt).  Plans of one thread replay in
// stream order anyway; sharing adds ordering only between replays that would otherwise overlap on a side stream.
struct SidePool {
    int device = -1;
    bool least_priority = false;
    std::vector<hipStream_t> streams;
    hipEvent_t wait_ev = nullptr;       // vhap_plan_side_stream_wait's event: one per pool, i.e. per (thread, device)
};
thread_local std::vector<SidePool> g_side_pools;
thread_local int g_side_base = 0;       // vhap_plan_set_side_base: the pool index of the NEXT plan's first side stream

SidePool* side_pool(bool least_priority) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (auto& sp : g_side_pools) if (sp.device == dev && sp.least_priority == least_priority) return &sp;
    g_side_pools.push_back(SidePool{dev, least_priority, {}, nullptr});
    return &g_side_pools.back();
}

hipStream_t pool_stream(int k, bool least_priority) {
    SidePool* pool = side_pool(least_priority);
    if (!pool) return nullptr;
    while ((int)pool->streams.size() <= k) {
        hipStream_t st = nullptr;
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        const hipError_t e = least_priority ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_least)
                                            : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) return nullptr;
        pool->streams.push_back(st);
    }
    return pool->streams[k];
}

void destroy(vhap_plan* p) {
    if (!p) return;
    for (auto e : p->events) if (e) (void)hipEventDestroy(e);
    for (auto e : p->tails) if (e) (void)hipEventDestroy(e);
    for (auto e : p->tev) if (e) (void)hipEventDestroy(e);
    if (p->start) (void)hipEventDestroy(p->start);
    // (the side streams belong to the thread's pool)
    delete p;
}

inline hipStream_t stream_of(const vhap_plan* p, int idx, hipStream_t launch) { return idx == 0 ? launch : p->streams[idx - 1]; }

hipError_t launch_node(const PlanNode& n, hipStream_t st) {
    if (n.type == 0)
        return hipLaunchKernel(n.kp.func, n.kp.gridDim, n.kp.blockDim, n.kp.kernelParams, n.kp.sharedMemBytes, st);
    if (n.type == 1) {
        const hipMemsetParams& m = n.ms;
        if (m.height <= 1) {
            if (m.elementSize == 4) return hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m.dst), (int)m.value, m.width, st);
            if (m.elementSize == 2) return hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(m.dst), (unsigned short)m.value, m.width, st);
            return hipMemsetAsync(m.dst, (int)(m.value & 0xff), m.width, st);
        }
        return hipMemset2DAsync(m.dst, m.pitch, (int)(m.value & 0xff), m.width * m.elementSize, m.height, st);   // (byte-uniform values only: checked at build)
    }
    if (n.type == 3) return hipMemcpyAsync(n.cp_dst, n.cp_src, n.cp_bytes, hipMemcpyDeviceToDevice, st);
    return hipSuccess;
}

}  // namespace

extern "C" int vhap_plan_from_graph(void* hip_graph, int max_streams, vhap_plan_t* out) {
    VHAP_ENTER();
    if (!hip_graph || !out) return VHAP_E_NULLPTR;
    if (max_streams < 1 || max_streams > 8) return VHAP_E_BADDIM;
    *out = nullptr;
    hipGraph_t g = static_cast<hipGraph_t>(hip_graph);
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return VHAP_E_HIP;
    if (n == 0 || n > 4096) return VHAP_E_BADDIM;
    std::vector<hipGraphNode_t> gn(n);
    if (hipGraphGetNodes(g, gn.data(), &n) != hipSuccess) return VHAP_E_HIP;
    auto index_of = [&](hipGraphNode_t x) {
        for (size_t i = 0; i < n; i++) if (gn[i] == x) return (int)i;
        return -1;
    };
    // ---- nodes and edges, in the graph's own (creation = capture) order ----
    std::vector<PlanNode> raw(n);
    for (size_t i = 0; i < n; i++) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(gn[i], &t) != hipSuccess) return VHAP_E_HIP;
        PlanNode& nd = raw[i];
        nd.raw = (int)i;
        nd.handle = gn[i];
        if (t == hipGraphNodeTypeKernel) {
            nd.type = 0;
            if (hipGraphKernelNodeGetParams(gn[i], &nd.kp) != hipSuccess) return VHAP_E_HIP;
            if (!nd.kp.func || (!nd.kp.kernelParams && nd.kp.extra)) return VHAP_E_UNSUPPORTED;   // (module launches with a packed argument buffer)
            const char* nm = hipKernelNameRefByPtr(nd.kp.func, nullptr);
            nd.name = nm ? nm : "?";
        } else if (t == hipGraphNodeTypeMemset) {
            nd.type = 1;
            if (hipGraphMemsetNodeGetParams(gn[i], &nd.ms) != hipSuccess) return VHAP_E_HIP;
            nd.name = "memset";
            // a 2-D memset is replayed through hipMemset2DAsync, which fills BYTES: 16- / 32-bit elements only with a byte-uniform value
            if (nd.ms.height > 1 && nd.ms.elementSize > 1) {
                const unsigned v = nd.ms.value, b0 = v & 0xffu;
                const bool uniform = nd.ms.elementSize == 2 ? ((v >> 8) & 0xffu) == b0
                                                            : (((v >> 8) & 0xffu) == b0 && ((v >> 16) & 0xffu) == b0 && ((v >> 24) & 0xffu) == b0);
                if (!uniform) {
                    fprintf(stderr, "vhap_plan_from_graph: node %zu is a 2-D memset of %u-byte elements with a non-uniform byte pattern\n", i, nd.ms.elementSize);
                    return VHAP_E_UNSUPPORTED;
                }
            }
        } else if (t == hipGraphNodeTypeMemcpy) {
            // same-layout copy_ / clone of the host framework inside a captured stage: a contiguous device-to-device copy, replayed as one
            hipMemcpy3DParms cp{};
            if (hipGraphMemcpyNodeGetParams(gn[i], &cp) != hipSuccess) return VHAP_E_HIP;
            const bool flat = cp.kind == hipMemcpyDeviceToDevice && !cp.srcArray && !cp.dstArray && cp.extent.height <= 1 && cp.extent.depth <= 1 &&
                              cp.srcPos.x == 0 && cp.srcPos.y == 0 && cp.srcPos.z == 0 && cp.dstPos.x == 0 && cp.dstPos.y == 0 && cp.dstPos.z == 0 &&
                              cp.srcPtr.ptr && cp.dstPtr.ptr;
            if (!flat) {
                fprintf(stderr, "vhap_plan_from_graph: node %zu is a memcpy node that is not a flat device-to-device copy\n", i);
                return VHAP_E_UNSUPPORTED;
            }
            nd.type = 3;
            nd.cp_dst = cp.dstPtr.ptr;
            nd.cp_src = cp.srcPtr.ptr;
            nd.cp_bytes = cp.extent.width;
            nd.name = "memcpy";
        } else if (t == hipGraphNodeTypeEmpty) {
            nd.type = 2;
            nd.name = "empty";
        } else {
            fprintf(stderr, "vhap_plan_from_graph: node %zu is a %s node (only kernel / memset / flat memcpy / empty nodes are supported)\n", i, node_type_name(t));
            return VHAP_E_UNSUPPORTED;
        }
        size_t nd_deps = 0;
        if (hipGraphNodeGetDependencies(gn[i], nullptr, &nd_deps) != hipSuccess) return VHAP_E_HIP;
        if (nd_deps) {
            std::vector<hipGraphNode_t> d(nd_deps);
            if (hipGraphNodeGetDependencies(gn[i], d.data(), &nd_deps) != hipSuccess) return VHAP_E_HIP;
            for (size_t k = 0; k < nd_deps; k++) {
                const int j = index_of(d[k]);
                if (j < 0) return VHAP_E_HIP;
                nd.deps.push_back(j);
            }
            std::sort(nd.deps.begin(), nd.deps.end());
        }
    }
    (void)hipGetLastError();
    // ---- launch order: Kahn's algorithm, always the lowest creation index among the ready nodes (the capture order, which IS a
    //      topological order, comes out unchanged; a runtime that lists nodes differently still gets a valid one) ----
    std::vector<int> indeg(n, 0), order, pos(n, -1);
    std::vector<std::vector<int>> succ(n);
    for (size_t i = 0; i < n; i++)
        for (int d : raw[i].deps) { succ[d].push_back((int)i); indeg[i]++; }
    std::vector<char> done(n, 0);
    for (size_t k = 0; k < n; k++) {
        int pick = -1;
        for (size_t i = 0; i < n; i++) if (!done[i] && indeg[i] == 0) { pick = (int)i; break; }
        if (pick < 0) return VHAP_E_HIP;                                         // a cycle: not a DAG
        done[pick] = 1;
        pos[pick] = (int)order.size();
        order.push_back(pick);
        for (int s : succ[pick]) indeg[s]--;
    }
    vhap_plan* p = new vhap_plan();
    (void)hipGetDevice(&p->device);
    p->nodes.resize(n);
    for (size_t k = 0; k < n; k++) {
        p->nodes[k] = raw[order[k]];
        for (int& d : p->nodes[k].deps) d = pos[d];
        std::sort(p->nodes[k].deps.begin(), p->nodes[k].deps.end());
    }
    // ---- ancestors (bit sets): a stream may take a node without a false dependency iff its tail is an ancestor of the node ----
    const size_t words = (n + 63) / 64;
    std::vector<uint64_t> anc(n * words, 0);
    for (size_t k = 0; k < n; k++)
        for (int d : p->nodes[k].deps) {
            anc[k * words + d / 64] |= 1ull << (d % 64);
            for (size_t w = 0; w < words; w++) anc[k * words + w] |= anc[(size_t)d * words + w];
        }
    auto is_anc = [&](int a, int k) { return (anc[(size_t)k * words + a / 64] >> (a % 64)) & 1ull; };
    // ---- stream assignment ----
    std::vector<int> tail(1, -1);                                                // tail[s] = last node placed on stream s (-1: none yet)
    for (size_t k = 0; k < n; k++) {
        PlanNode& nd = p->nodes[k];
        int s = -1;
        // 1. continue a chain: a predecessor that is still the tail of its stream (the main stream first, then the lowest stream)
        for (int d : nd.deps) {
            const int ds = p->nodes[d].stream;
            if (tail[ds] == d && (s < 0 || ds < s)) s = ds;
        }
        // 2. a stream whose tail is an ancestor (its work is done before this node may start anyway) -- the launch stream only for
        //    the very first node: side work parked there would sit in front of the main chain's next kernel
        if (s < 0 && tail[0] < 0) s = 0;
        if (s < 0)
            for (size_t c = 1; c < tail.size(); c++)
                if (tail[c] < 0 || is_anc(tail[c], (int)k)) { s = (int)c; break; }
        // 3. a new stream, or (at the limit) the side stream whose tail is oldest
        if (s < 0) {
            if ((int)tail.size() < max_streams) {
                tail.push_back(-1);
                s = (int)tail.size() - 1;
            } else if (tail.size() == 1) {
                s = 0;
            } else {
                s = 1;
                for (size_t c = 2; c < tail.size(); c++) if (tail[c] < tail[s]) s = (int)c;
            }
        }
        nd.stream = s;
        tail[s] = (int)k;
    }
    const int ns = (int)tail.size();
    // ---- cross-stream edges -> events; an edge is dropped when the consumer's stream already waited for a LATER node of the
    //      producer's stream (streams are in order) ----
    std::vector<int> waited((size_t)ns * ns, -1);                                // waited[to * ns + from] = latest node of `from` that `to` waits for
    for (size_t k = 0; k < n; k++) {
        PlanNode& nd = p->nodes[k];
        for (int d : nd.deps) {
            const int from = p->nodes[d].stream, to = nd.stream;
            if (from == to) continue;
            if (waited[(size_t)to * ns + from] >= d) continue;
            waited[(size_t)to * ns + from] = d;
            if (p->nodes[d].record < 0) {
                p->nodes[d].record = (int)p->events.size();
                p->events.push_back(nullptr);
            }
            nd.waits.push_back(p->nodes[d].record);
        }
    }
    // ---- open tails: side-stream nodes that nothing on stream 0 waits for, directly, through a successor, or through a LATER node of
    //      their own stream -- what a DEFER_JOIN replay leaves running when stream 0's last node has finished ----
    {
        std::vector<char> covered(n, 0), later(ns, 0);
        for (size_t k = n; k-- > 0;) {
            PlanNode& nd = p->nodes[k];
            if (nd.stream == 0 || later[nd.stream]) covered[k] = 1;
            if (covered[k]) {
                later[nd.stream] = 1;
                for (int d : nd.deps) covered[d] = 1;             // (deps precede k in launch order: visited later in this reverse sweep)
            }
        }
        // a dependency marked covered above may sit on a side stream: everything before it on that stream is covered too
        std::fill(later.begin(), later.end(), 0);
        for (size_t k = n; k-- > 0;) {
            PlanNode& nd = p->nodes[k];
            if (covered[k]) later[nd.stream] = 1;
            else if (later[nd.stream]) covered[k] = 1;
            nd.open_tail = !covered[k];
        }
    }
    // Edge events order kernels of ONE device: no system-scope fence (hipEventDisableSystemFence).  A default event makes its record a
    // cache write-back / invalidate towards the host, and the next kernel of the recording stream waits for it: ~8 us at every one of the
    // ~10 records on the step's main chain.  The tail events (what a host-side synchronisation of the launch stream eventually stands on)
    // keep the system fence.  (debug flag 65536: A/B, every event with the fence)
    const unsigned edge_flags = hipEventDisableTiming | ((vhap_g_debug_flags & 65536) ? 0u : (unsigned)hipEventDisableSystemFence);
    bool ok = true;
    for (auto& e : p->events) ok = ok && hipEventCreateWithFlags(&e, edge_flags) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&p->start, edge_flags) == hipSuccess;
    p->streams.assign(ns - 1, nullptr);
    p->tails.assign(ns - 1, nullptr);
    for (int s = 0; s + 1 < ns && ok; s++) {
        // (debug flag 524288: A/B, the side streams at the LOWEST priority -- the launch stream carries the dependency chain of the step)
        p->streams[s] = pool_stream(g_side_base + s, (vhap_g_debug_flags & 524288) != 0);
        ok = ok && p->streams[s] != nullptr;
        ok = ok && hipEventCreateWithFlags(&p->tails[s], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { destroy(p); return VHAP_E_HIP; }
    *out = p;
    return VHAP_OK;
}

// Which pool streams the NEXT plan created on this thread takes: its side stream k = pool stream base + k.  Two plans replayed back to
// back whose side chains must overlap (the sharded step's pixel plan leaves its texture chain running under the geometry plan) take
// different ones.  vhap_plan_touch_side_streams: a 4-byte fill on pool streams 0 .. n - 1, in order -- HIP binds a stream to one of
// its four hardware queues at the stream's FIRST command, round-robin; a host that touches its launch stream, these and its communication
// stream one after the other gets them onto four different queues.
extern "C" int vhap_plan_set_side_base(int base) {
    if (base < 0 || base > 8) return VHAP_E_BADDIM;
    g_side_base = base;
    return VHAP_OK;
}

// Pool stream k of this thread waits for everything enqueued on `other` so far -- and nothing else does.  The sharded step uses it at the
// head of a replay: only the forward plan's texture chain (its first side stream) needs the all-gathered texture, the geometry head on the
// launch stream starts under the transfer.
extern "C" int vhap_plan_side_stream_wait(int k, vhap_stream_t other) {
    VHAP_ENTER();
    if (k < 0 || k > 8) return VHAP_E_BADDIM;
    const bool lp = (vhap_g_debug_flags & 524288) != 0;
    hipStream_t st = pool_stream(k, lp);
    SidePool* pool = side_pool(lp);
    if (!st || !pool) return VHAP_E_HIP;
    // (the event belongs to the pool of the CURRENT device: a thread that drives two devices must not record one device's event on the
    // other's stream -- round-5 advisor)
    if (!pool->wait_ev && hipEventCreateWithFlags(&pool->wait_ev, hipEventDisableTiming) != hipSuccess) return VHAP_E_HIP;
    if (hipEventRecord(pool->wait_ev, vhap_stream(other)) != hipSuccess) return VHAP_E_HIP;
    if (hipStreamWaitEvent(st, pool->wait_ev, 0) != hipSuccess) return VHAP_E_HIP;
    return VHAP_OK;
}

// The calling thread's side-stream pools (every device), destroyed: for short-lived worker threads that created plans -- the pools otherwise
// live as long as the thread's storage and their HIP streams are never returned.  No plan of this thread may be replayed afterwards
// (plans hold the pool's stream handles); create new plans instead.  Synchronises each stream first.
extern "C" int vhap_plan_pool_release(void) {
    VHAP_ENTER();
    int rc = VHAP_OK;
    for (auto& sp : g_side_pools) {
        for (auto st : sp.streams) {
            if (hipStreamSynchronize(st) != hipSuccess) rc = VHAP_E_HIP;
            if (hipStreamDestroy(st) != hipSuccess) rc = VHAP_E_HIP;
        }
        if (sp.wait_ev && hipEventDestroy(sp.wait_ev) != hipSuccess) rc = VHAP_E_HIP;
    }
    g_side_pools.clear();
    (void)hipGetLastError();
    return rc;
}

extern "C" int vhap_plan_touch_side_streams(int n, void* scratch_4_bytes) {
    VHAP_ENTER();
    if (!scratch_4_bytes) return VHAP_E_NULLPTR;
    if (n < 0 || n > 8) return VHAP_E_BADDIM;
    for (int k = 0; k < n; k++) {
        hipStream_t st = pool_stream(k, (vhap_g_debug_flags & 524288) != 0);
        if (!st) return VHAP_E_HIP;
        vhap_zero_async(scratch_4_bytes, 4, st);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

extern "C" int vhap_plan_destroy(vhap_plan_t plan) {
    VHAP_ENTER();
    if (!plan) return VHAP_E_NULLPTR;
    destroy(plan);
    return VHAP_OK;
}

extern "C" int vhap_plan_info(vhap_plan_t plan, int* n_nodes, int* n_streams, int* n_events) {
    if (!plan) return VHAP_E_NULLPTR;
    if (n_nodes) *n_nodes = (int)plan->nodes.size();
    if (n_streams) *n_streams = (int)plan->streams.size() + 1;
    if (n_events) *n_events = (int)plan->events.size();
    return VHAP_OK;
}

// One line per node: "index stream name <- deps | waits | record" (for DESIGN.md / debugging); returns the length needed.
extern "C" size_t vhap_plan_describe(vhap_plan_t plan, char* buf, size_t cap) {
    if (!plan) return 0;
    std::string s;
    char line[512];
    for (size_t k = 0; k < plan->nodes.size(); k++) {
        const PlanNode& nd = plan->nodes[k];
        std::string nm = nd.name.substr(0, nd.name.find('('));
        snprintf(line, sizeof line, "%3zu s%d %-48.48s <-", k, nd.stream, nm.c_str());
        s += line;
        for (int d : nd.deps) { snprintf(line, sizeof line, " %d", d); s += line; }
        if (!nd.waits.empty()) { s += " | waits"; for (int e : nd.waits) { snprintf(line, sizeof line, " e%d", e); s += line; } }
        if (nd.record >= 0) { snprintf(line, sizeof line, " | records e%d", nd.record); s += line; }
        s += "\n";
    }
    if (buf && cap) {
        const size_t m = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return s.size() + 1;
}

extern "C" int vhap_plan_node_name(vhap_plan_t plan, int node, char* buf, size_t cap) {
    if (!plan || !buf || !cap) return VHAP_E_NULLPTR;
    if (node < 0 || node >= (int)plan->nodes.size()) return VHAP_E_BADDIM;
    const std::string& nm = plan->nodes[node].name;
    const size_t m = std::min(cap - 1, nm.size());
    memcpy(buf, nm.data(), m);
    buf[m] = 0;
    return VHAP_OK;
}

// an error in the middle of a replay: whatever was enqueued on the side streams is joined back into the launch stream (best effort), so that
// the caller's stream order covers everything that was issued, and the plan reports no open tails
static int plan_abort(vhap_plan* p, hipStream_t launch) {
    (void)hipGetLastError();
    for (size_t s = 0; s < p->streams.size(); s++)
        if (hipEventRecord(p->tails[s], p->streams[s]) == hipSuccess) (void)hipStreamWaitEvent(launch, p->tails[s], 0);
    p->tails_open = false;
    return VHAP_E_HIP;
}

static int plan_launch(vhap_plan* p, hipStream_t launch, bool timed, bool defer_join = false) {
    const size_t n = p->nodes.size();
#define PLAN_HIP(x) do { if ((x) != hipSuccess) return plan_abort(p, launch); } while (0)
    if (timed && p->tev.empty()) {
        p->tev.assign(2 * n + 1, nullptr);
        for (auto& e : p->tev) PLAN_HIP(hipEventCreate(&e));
    }
    if (!p->streams.empty() || timed) {
        PLAN_HIP(hipEventRecord(timed ? p->tev[2 * n] : p->start, launch));
        for (auto s : p->streams) PLAN_HIP(hipStreamWaitEvent(s, timed ? p->tev[2 * n] : p->start, 0));
    }
    for (size_t k = 0; k < n; k++) {
        const PlanNode& nd = p->nodes[k];
        hipStream_t st = stream_of(p, nd.stream, launch);
        for (int e : nd.waits) PLAN_HIP(hipStreamWaitEvent(st, p->events[e], 0));
        if (timed) PLAN_HIP(hipEventRecord(p->tev[2 * k], st));
        // An edge event recorded by a SEPARATE packet behind the kernel costs the recording stream ~5 us before its next kernel starts (the
        // step's main chain records four).  A kernel node hands its edge event to the dispatch itself (hipExtLaunchKernel's stop event: the
        // packet's own completion signal) -- debug flag 2097152: A/B, the separate record.
        const bool bound = nd.record >= 0 && nd.type == 0 && !timed && !(vhap_g_debug_flags & 2097152);
        if (bound)
            PLAN_HIP(hipExtLaunchKernel(nd.kp.func, nd.kp.gridDim, nd.kp.blockDim, nd.kp.kernelParams, nd.kp.sharedMemBytes, st, nullptr,
                                        p->events[nd.record], 0));
        else
            PLAN_HIP(launch_node(nd, st));
        if (timed) PLAN_HIP(hipEventRecord(p->tev[2 * k + 1], st));
        if (nd.record >= 0 && !bound) PLAN_HIP(hipEventRecord(p->events[nd.record], st));
    }
    // The tail events: a joined replay records and waits here; a replay that defers its join leaves the record to vhap_plan_join (in-order
    // streams: an event recorded later covers the same work) -- one packet less per side stream and replay, and none between the open tail
    // and the next replay's first node on its stream.  (debug flag 4194304: A/B, record here in both cases)
    const bool lazy_tails = defer_join && !(vhap_g_debug_flags & 4194304);
    for (size_t s = 0; s < p->streams.size() && !lazy_tails; s++) {
        PLAN_HIP(hipEventRecord(p->tails[s], p->streams[s]));
        if (!defer_join) PLAN_HIP(hipStreamWaitEvent(launch, p->tails[s], 0));
    }
    p->tails_open = defer_join && !p->streams.empty();
    p->tails_recorded = !lazy_tails;
#undef PLAN_HIP
    return VHAP_OK;
}

extern "C" int vhap_plan_launch(vhap_plan_t plan, vhap_stream_t stream, int call_flags) {
    VHAP_ENTER();
    if (!plan) return VHAP_E_NULLPTR;
    return plan_launch(plan, vhap_stream(stream), false, (call_flags & VHAP_CALL_PLAN_DEFER_JOIN) != 0);
}

// `stream` waits for the side-stream work of the last replay (no-op when that replay joined itself)
extern "C" int vhap_plan_join(vhap_plan_t plan, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!plan) return VHAP_E_NULLPTR;
    if (plan->tails_open) {
        for (size_t s = 0; s < plan->streams.size(); s++) {
            if (!plan->tails_recorded && hipEventRecord(plan->tails[s], plan->streams[s]) != hipSuccess) return VHAP_E_HIP;
            if (hipStreamWaitEvent(vhap_stream(stream), plan->tails[s], 0) != hipSuccess) return VHAP_E_HIP;
        }
        plan->tails_open = false;
        plan->tails_recorded = true;
    }
    return VHAP_OK;
}

// nodes of the NEXT replay that are not ordered behind the open tails of a DEFER_JOIN replay.  Stream order places a node only behind the
// tails of its OWN stream: with open tails on several side streams a node is ordered iff, for EVERY stream that carries an open tail, it
// sits on that stream or (transitively, through dependencies or the stream order inside the replay) follows a node that does.
extern "C" int vhap_plan_free_heads(vhap_plan_t plan, int* nodes, int cap) {
    if (!plan) return VHAP_E_NULLPTR;
    const size_t n = plan->nodes.size();
    const int ns = (int)plan->streams.size() + 1;
    unsigned tail_mask = 0u;                                   // streams with an open tail (ns <= 8)
    for (const PlanNode& nd : plan->nodes) if (nd.open_tail) tail_mask |= 1u << nd.stream;
    std::vector<unsigned> behind(n, 0u), last_on(ns, 0u);      // behind[k]: streams node k is ordered behind; last_on[s]: the same for stream s's latest node
    int m = 0;
    for (size_t k = 0; k < n; k++) {
        const PlanNode& nd = plan->nodes[k];
        unsigned b = (1u << nd.stream) | last_on[nd.stream];   // its own stream, and whatever its stream predecessor already follows
        for (int d : nd.deps) b |= behind[d];
        behind[k] = b;
        last_on[nd.stream] = b;
        if ((b & tail_mask) != tail_mask) {
            if (nodes && m < cap) nodes[m] = (int)k;
            m++;
        }
    }
    return m;
}

// the captured graph's node behind plan node k (identity only), and the nodes the graph `stream` is capturing into holds right now (at most
// `cap` handles written; returns their number, 0: not capturing): a host that reads the latter after every call of a capture knows which
// call created which node -- the order of a graph's node list is the runtime's business -- hence which buffers a plan node touches
extern "C" void* vhap_plan_node_handle(vhap_plan_t plan, int node) {
    if (!plan || node < 0 || node >= (int)plan->nodes.size()) return nullptr;
    return plan->nodes[node].handle;
}

extern "C" int vhap_capture_nodes(vhap_stream_t stream, void** nodes, int cap) {
    VHAP_ENTER();
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    hipGraph_t g = nullptr;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo_v2(vhap_stream(stream), &status, &id, &g, nullptr, nullptr) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (status != hipStreamCaptureStatusActive || !g) return 0;
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (nodes && cap > 0 && n > 0) {
        std::vector<hipGraphNode_t> gn(n);
        if (hipGraphGetNodes(g, gn.data(), &n) != hipSuccess) { (void)hipGetLastError(); return 0; }
        for (size_t i = 0; i < n && (int)i < cap; i++) nodes[i] = gn[i];
    }
    return (int)n;
}

// nodes a DEFER_JOIN replay leaves un-joined: indices into launch order, at most `cap` written; returns their number
extern "C" int vhap_plan_open_tails(vhap_plan_t plan, int* nodes, int cap) {
    if (!plan) return VHAP_E_NULLPTR;
    int m = 0;
    for (size_t k = 0; k < plan->nodes.size(); k++)
        if (plan->nodes[k].open_tail) {
            if (nodes && m < cap) nodes[m] = (int)k;
            m++;
        }
    return m;
}

// One replay with every node bracketed by timing events; BLOCKS until it has finished.  start_us[k] = start of node k relative to the
// head of the replay, dur_us[k] = its duration (both in microseconds; the brackets cost a few microseconds per node, so the replay as a
// whole is slower than an untimed one -- use it for per-kernel numbers, not for the step time).
extern "C" int vhap_plan_launch_timed(vhap_plan_t plan, vhap_stream_t stream, float* start_us, float* dur_us, int n) {
    VHAP_ENTER();
    if (!plan || !start_us || !dur_us) return VHAP_E_NULLPTR;
    if (n < (int)plan->nodes.size()) return VHAP_E_BADDIM;
    const int rc = plan_launch(plan, vhap_stream(stream), true);
    if (rc != VHAP_OK) return rc;
    if (hipStreamSynchronize(vhap_stream(stream)) != hipSuccess) return VHAP_E_HIP;
    const size_t nn = plan->nodes.size();
    for (size_t k = 0; k < nn; k++) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, plan->tev[2 * nn], plan->tev[2 * k]) != hipSuccess) return VHAP_E_HIP;
        if (hipEventElapsedTime(&b, plan->tev[2 * k], plan->tev[2 * k + 1]) != hipSuccess) return VHAP_E_HIP;
        start_us[k] = a * 1000.f;
        dur_us[k] = b * 1000.f;
    }
    return VHAP_OK;
}
