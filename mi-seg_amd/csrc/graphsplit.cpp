// A captured multi-stream hipGraph replayed as single-stream graphs (ABI 7: miseg_graph_split_*).  Opt-in (MISEG_GRAPH_SPLIT=1): an
// instrument and an alternative replay, not the default - the numbers are in DESIGN.md R4.3.
//
// Why it was built.  On MI355X / ROCm 7 a graph that is ONE chain of small kernels replays at 1.6 us per dependent kernel; the same chain
// beside ONE 80 us kernel on a side stream, as one captured graph, at 3.1 us per kernel (scripts/debug/graph_edges_probe.py, device time
// behind a blocker: 400 kernels 700 -> 1250 us; additive for longer kernels, 200 x 2.9 us: 585 -> 884 us).  Crossing edges are not the cost
// (16 more edges: +5 %).  What is: a stream that SITS ON A WAIT for an event of another stream costs the streams that are running ~1.3 us on
// every kernel (scripts/debug/graph_two_queue_probe2.py: two single-chain graphs on two streams overlap at full speed, 780 us, until a third
// stream - or the launch stream, which is what the runtime's replay of a multi-stream graph leaves waiting for its internal streams - waits
// for their end: 1280 us).
//
// What it does.  The captured graph is decomposed into chains ("lanes": repeated longest path - the main stream's nodes, then each stretch
// of a side stream), every lane is cut where an edge crosses to or from another lane, every piece becomes a graph of its own (a clone of
// the captured graph with all other nodes destroyed: kernel arguments, copies and memsets are carried over by the runtime, nothing is
// re-created here) and is instantiated.  A replay launches the pieces in topological order - the longest lane on the CALLER's stream (so
// that nothing waits for it), the other lanes on streams the plan owns, chosen by experiment so that they really run concurrently (streams
// share the process's 4 hardware queues) - with an event per crossing edge: the same partial order as the captured graph.
//
// What it gives.  The probe's shape: 1250 -> 790 us.  The training step (354 nodes, 3 lanes, 11 pieces): 149 patches/s against 153 with
// the runtime's replay - its kernels are not 2 us long, every piece boundary costs, and the side stream still waits 2.5 ms for the fork of
// the backward pass.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "../../include/miseg_hip_debug.h"

namespace {

struct Segment {
  int lane = 0, stream = 0;              // stream 0 = the caller's
  std::vector<int> nodes;                // topological indices, ascending
  std::vector<int> waits;                // segments whose `done` event this one waits for (other streams only)
  bool record = false;                   // some other stream waits for this segment
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipEvent_t done = nullptr;
};

struct Plan {
  std::vector<Segment> segs;             // launch order
  std::vector<hipStream_t> side;         // streams 1 .. of the plan
  std::vector<hipEvent_t> side_end;
  hipEvent_t start = nullptr;
  hipEvent_t tail = nullptr;             // recorded on the caller's stream behind the last piece of a launch: destroy waits for it
  bool launched = false;
  int nodes = 0, lanes = 0, cross = 0;
  bool concurrent = true;                 // the side streams were seen to run beside the launch stream and each other
};

#define GS_CHECK(call)                                                                                              \
  do {                                                                                                              \
    hipError_t e__ = (call);                                                                                        \
    if (e__ != hipSuccess) return miseg::set_error(MISEG_E_LAUNCH, "graph_split: %s: %s", #call, hipGetErrorString(e__)); \
  } while (0)

// Two HIP streams may sit on ONE hardware queue (GPU_MAX_HW_QUEUES = 4 queues are shared by all streams of the process, a new stream takes
// the least used one - two streams created back to back can both land on it): pieces launched on them would run one after the other.
// The plan therefore picks its streams by experiment: a 100 us spin kernel on each of two candidates, started together - ~105 us when
// they overlap, >= 200 us when they share a queue.
__global__ void gs_spin_kernel(unsigned long long ticks) {      // wall_clock64: 100 MHz
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

bool run_concurrently(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1, hipEvent_t eb) {
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {      // (rep 0 warms up: the first launch loads the code object)
    if (hipEventRecord(e0, a) != hipSuccess || hipStreamWaitEvent(b, e0, 0) != hipSuccess) return false;
    gs_spin_kernel<<<1, 64, 0, a>>>(10000);
    gs_spin_kernel<<<1, 64, 0, b>>>(10000);
    if (hipEventRecord(eb, b) != hipSuccess || hipStreamWaitEvent(a, eb, 0) != hipSuccess || hipEventRecord(e1, a) != hipSuccess) return false;
    if (hipEventSynchronize(e1) != hipSuccess) return false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
    if (rep) best = ms < best ? ms : best;
  }
  return best < 0.16f;      // 100 us each: together ~0.105, one after the other >= 0.2
}

// k fresh streams that overlap pairwise and with `with`, out of up to `tries` candidates (the rest are destroyed); fewer than k when the
// search fails
std::vector<hipStream_t> concurrent_streams(hipStream_t with, int k, int tries = 12) {
  std::vector<hipStream_t> got, reject;
  hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) return got;
  for (int t = 0; t < tries && (int)got.size() < k; ++t) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
    bool ok = run_concurrently(with, s, e0, e1, eb);
    for (hipStream_t g : got)
      if (ok && !run_concurrently(g, s, e0, e1, eb)) ok = false;
    (ok ? got : reject).push_back(s);
  }
  for (hipStream_t s : reject) (void)hipStreamDestroy(s);      // (destroyed only now: a destroyed stream's queue slot would be handed out again)
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eb);
  return got;
}

void destroy_plan(Plan* p) {
  if (!p) return;
  // pieces of a launch may still be in flight: their execs, events and side streams are freed below
  if (p->launched && p->tail) (void)hipEventSynchronize(p->tail);
  for (auto s : p->side)
    if (s) (void)hipStreamSynchronize(s);
  for (auto& s : p->segs) {
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
    if (s.graph) (void)hipGraphDestroy(s.graph);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (auto e : p->side_end) (void)hipEventDestroy(e);
  for (auto s : p->side)
    if (s) (void)hipStreamDestroy(s);
  if (p->start) (void)hipEventDestroy(p->start);
  if (p->tail) (void)hipEventDestroy(p->tail);
  delete p;
}

}  // namespace

extern "C" int miseg_graph_split_create(void* graph_, miseg_stream_t launch_stream, int max_side_streams, miseg_graph_split_t* out, miseg_graph_split_info* info) {
  MISEG_REQUIRE(graph_ && out, MISEG_E_BADARG, "graph_split_create: null argument");
  hipGraph_t graph = (hipGraph_t)graph_;
  *out = nullptr;
  size_t n = 0, ne = 0;
  GS_CHECK(hipGraphGetNodes(graph, nullptr, &n));
  MISEG_REQUIRE(n > 0, MISEG_E_BADARG, "graph_split_create: empty graph");
  std::vector<hipGraphNode_t> nodes(n);
  GS_CHECK(hipGraphGetNodes(graph, nodes.data(), &n));
  GS_CHECK(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
  std::vector<hipGraphNode_t> ef(ne), et(ne);
  if (ne) GS_CHECK(hipGraphGetEdges(graph, ef.data(), et.data(), &ne));
  // node handle -> index
  std::vector<std::pair<hipGraphNode_t, int>> byptr(n);
  for (size_t i = 0; i < n; ++i) byptr[i] = {nodes[i], (int)i};
  std::sort(byptr.begin(), byptr.end());
  auto idx_of = [&](hipGraphNode_t h) {
    auto it = std::lower_bound(byptr.begin(), byptr.end(), std::make_pair(h, -1));
    return (it != byptr.end() && it->first == h) ? it->second : -1;
  };
  std::vector<std::vector<int>> succ(n), pred(n);
  for (size_t e = 0; e < ne; ++e) {
    const int a = idx_of(ef[e]), b = idx_of(et[e]);
    MISEG_REQUIRE(a >= 0 && b >= 0, MISEG_E_UNSUPPORTED, "graph_split_create: an edge names a node the graph does not list");
    succ[a].push_back(b);
    pred[b].push_back(a);
  }
  // topological order (Kahn; ties by the runtime's node order, which is creation order)
  std::vector<int> topo, indeg(n), pos(n, -1);
  for (size_t i = 0; i < n; ++i) indeg[i] = (int)pred[i].size();
  {
    std::vector<int> ready;
    for (int i = (int)n - 1; i >= 0; --i)
      if (!indeg[i]) ready.push_back(i);
    while (!ready.empty()) {
      const int v = ready.back();
      ready.pop_back();
      pos[v] = (int)topo.size();
      topo.push_back(v);
      for (int w : succ[v])
        if (--indeg[w] == 0) ready.push_back(w);
      std::sort(ready.begin(), ready.end(), [](int a, int b) { return a > b; });
    }
  }
  MISEG_REQUIRE(topo.size() == n, MISEG_E_UNSUPPORTED, "graph_split_create: the graph has a cycle");
  // lanes: repeated longest path among the nodes not yet taken
  std::vector<int> lane(n, -1);
  std::vector<std::vector<int>> lanes;      // node indices in topological order
  {
    size_t left = n;
    std::vector<int> len(n), from(n);
    while (left) {
      int best = -1;
      for (int v : topo) {
        if (lane[v] >= 0) continue;
        len[v] = 1;
        from[v] = -1;
        for (int p : pred[v])
          if (lane[p] < 0 && len[p] + 1 > len[v]) { len[v] = len[p] + 1; from[v] = p; }
        if (best < 0 || len[v] > len[best]) best = v;
      }
      std::vector<int> path;
      for (int v = best; v >= 0; v = from[v]) path.push_back(v);
      std::reverse(path.begin(), path.end());
      for (int v : path) lane[v] = (int)lanes.size();
      left -= path.size();
      lanes.push_back(path);
    }
  }
  const int nl = (int)lanes.size();
  std::vector<int> lpos(n);      // position inside its lane
  for (auto& l : lanes)
    for (size_t i = 0; i < l.size(); ++i) lpos[l[i]] = (int)i;
  // crossing edges; dominated ones dropped (u -> v is implied when a later node of u's lane reaches an earlier-or-equal node of v's lane)
  struct Cross { int u, v; };
  std::vector<Cross> cross;
  for (size_t a = 0; a < n; ++a)
    for (int b : succ[a])
      if (lane[a] != lane[b]) cross.push_back({(int)a, b});
  {
    std::vector<Cross> keep;
    for (size_t i = 0; i < cross.size(); ++i) {
      bool dom = false;
      for (size_t j = 0; j < cross.size() && !dom; ++j) {
        if (i == j) continue;
        const Cross &c = cross[i], &d = cross[j];
        if (lane[c.u] != lane[d.u] || lane[c.v] != lane[d.v]) continue;
        if (lpos[d.u] >= lpos[c.u] && lpos[d.v] <= lpos[c.v] && (lpos[d.u] != lpos[c.u] || lpos[d.v] != lpos[c.v] || j < i)) dom = true;
      }
      if (!dom) keep.push_back(cross[i]);
    }
    cross.swap(keep);
  }
  // cuts: a lane is cut behind every source and in front of every target of a crossing edge
  std::vector<char> cut_after(n, 0), cut_before(n, 0);
  for (auto& c : cross) { cut_after[c.u] = 1; cut_before[c.v] = 1; }
  Plan* plan = new Plan;
  plan->nodes = (int)n;
  plan->lanes = nl;
  plan->cross = (int)cross.size();
  std::vector<int> seg_of(n, -1);
  for (int l = 0; l < nl; ++l) {
    Segment cur;
    cur.lane = l;
    for (size_t i = 0; i < lanes[l].size(); ++i) {
      const int v = lanes[l][i];
      if (!cur.nodes.empty() && cut_before[v]) { plan->segs.push_back(cur); cur.nodes.clear(); }
      cur.nodes.push_back(v);
      if (cut_after[v]) { plan->segs.push_back(cur); cur.nodes.clear(); }
    }
    if (!cur.nodes.empty()) plan->segs.push_back(cur);
  }
  // launch order: by the topological position of a segment's first node (a crossing edge u -> v has pos(first(seg(u))) <= pos(u) < pos(v) =
  // pos(first(seg(v))), so the record of an event is always issued before its wait)
  std::sort(plan->segs.begin(), plan->segs.end(), [&](const Segment& a, const Segment& b) { return pos[a.nodes[0]] < pos[b.nodes[0]]; });
  for (size_t s = 0; s < plan->segs.size(); ++s)
    for (int v : plan->segs[s].nodes) seg_of[v] = (int)s;
  // lanes -> streams: lane 0 (the longest chain) on the caller's stream; another lane shares a side stream with an earlier lane only when
  // that lane's last node is an ancestor of its first node (the order the stream adds is then already a dependency)
  std::vector<int> lane_stream(nl, 0);
  {
    std::vector<std::vector<int>> on_stream;      // side stream -> lanes
    std::vector<char> reach(n);
    auto ancestor = [&](int a, int b) {            // is a an ancestor of b
      if (pos[a] >= pos[b]) return false;
      std::fill(reach.begin(), reach.end(), 0);
      std::vector<int> st{a};
      reach[a] = 1;
      while (!st.empty()) {
        const int v = st.back();
        st.pop_back();
        if (v == b) return true;
        for (int w : succ[v])
          if (!reach[w] && pos[w] <= pos[b]) { reach[w] = 1; st.push_back(w); }
      }
      return false;
    };
    std::vector<int> order;
    for (int l = 1; l < nl; ++l) order.push_back(l);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return pos[lanes[a][0]] < pos[lanes[b][0]]; });
    for (int l : order) {
      int got = -1;
      for (size_t s = 0; s < on_stream.size() && got < 0; ++s)
        if (ancestor(lanes[on_stream[s].back()].back(), lanes[l][0])) got = (int)s;
      if (got < 0) {
        if ((int)on_stream.size() >= max_side_streams) {
          destroy_plan(plan);
          return miseg::set_error(MISEG_E_UNSUPPORTED, "graph_split_create: the graph needs more than %d side stream(s)", max_side_streams);
        }
        on_stream.emplace_back();
        got = (int)on_stream.size() - 1;
      }
      on_stream[got].push_back(l);
      lane_stream[l] = got + 1;
    }
    plan->side.resize(on_stream.size(), nullptr);
    plan->side_end.resize(on_stream.size(), nullptr);
  }
  for (auto& s : plan->segs) s.stream = lane_stream[s.lane];
  for (auto& c : cross) {
    Segment &a = plan->segs[seg_of[c.u]], &b = plan->segs[seg_of[c.v]];
    if (a.stream == b.stream) continue;      // stream order (the launch order respects the dependency)
    a.record = true;
    if (std::find(b.waits.begin(), b.waits.end(), seg_of[c.u]) == b.waits.end()) b.waits.push_back(seg_of[c.u]);
  }
  // the pieces: clone, destroy every node of the clone that is not in the segment, instantiate
  std::vector<char> in_seg(n);
  for (auto& s : plan->segs) {
    hipError_t e = hipGraphClone(&s.graph, graph);
    if (e != hipSuccess) { destroy_plan(plan); return miseg::set_error(MISEG_E_LAUNCH, "graph_split_create: hipGraphClone: %s", hipGetErrorString(e)); }
    std::fill(in_seg.begin(), in_seg.end(), 0);
    for (int v : s.nodes) in_seg[v] = 1;
    for (size_t i = 0; i < n; ++i) {
      if (in_seg[i]) continue;
      hipGraphNode_t cn = nullptr;
      e = hipGraphNodeFindInClone(&cn, nodes[i], s.graph);
      if (e == hipSuccess) e = hipGraphDestroyNode(cn);
      if (e != hipSuccess) { destroy_plan(plan); return miseg::set_error(MISEG_E_LAUNCH, "graph_split_create: removing a node from a clone: %s", hipGetErrorString(e)); }
    }
    e = hipGraphInstantiate(&s.exec, s.graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { destroy_plan(plan); return miseg::set_error(MISEG_E_LAUNCH, "graph_split_create: hipGraphInstantiate: %s", hipGetErrorString(e)); }
    if (s.record) {
      e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
      if (e != hipSuccess) { destroy_plan(plan); return miseg::set_error(MISEG_E_LAUNCH, "graph_split_create: hipEventCreate: %s", hipGetErrorString(e)); }
    }
  }
  hipError_t e = hipEventCreateWithFlags(&plan->start, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&plan->tail, hipEventDisableTiming);
  for (size_t i = 0; i < plan->side.size() && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&plan->side_end[i], hipEventDisableTiming);
  if (e != hipSuccess) { destroy_plan(plan); return miseg::set_error(MISEG_E_LAUNCH, "graph_split_create: events: %s", hipGetErrorString(e)); }
  {
    std::vector<hipStream_t> st = concurrent_streams((hipStream_t)launch_stream, (int)plan->side.size());
    plan->concurrent = st.size() == plan->side.size();
    while (st.size() < plan->side.size()) {      // no such streams found: the replay stays correct, pieces may queue behind each other
      hipStream_t s_ = nullptr;
      if (hipStreamCreateWithFlags(&s_, hipStreamNonBlocking) != hipSuccess) {
        for (hipStream_t t_ : st) (void)hipStreamDestroy(t_);
        destroy_plan(plan);
        return miseg::set_error(MISEG_E_LAUNCH, "graph_split_create: hipStreamCreate failed");
      }
      st.push_back(s_);
    }
    for (size_t i = 0; i < plan->side.size(); ++i) plan->side[i] = st[i];
  }
  if (info) {
    info->nodes = plan->nodes;
    info->lanes = plan->lanes;
    info->segments = (int)plan->segs.size();
    info->crossing_edges = plan->cross;
    info->side_streams = (int)plan->side.size();
    info->main_lane_nodes = (int)lanes[0].size();
    info->streams_concurrent = plan->concurrent ? 1 : 0;
  }
  if (false) {      // (debug listing of the pieces; was MISEG_DEBUG_GRAPH_SPLIT: the library reads no environment)
    fprintf(stderr, "[graph_split] %d nodes, %zu edges, %d lanes, %d crossing edges, %zu segments, %zu side stream(s)%s\n", (int)n, ne, nl, plan->cross,
            plan->segs.size(), plan->side.size(), plan->concurrent ? "" : " (NOT seen to run concurrently)");
    for (size_t s = 0; s < plan->segs.size(); ++s) {
      const Segment& g = plan->segs[s];
      int nk = 0, nc = 0, nm = 0, no = 0;      // kernel / memcpy / memset / other nodes
      for (int v : g.nodes) {
        hipGraphNodeType ty = hipGraphNodeTypeEmpty;
        (void)hipGraphNodeGetType(nodes[v], &ty);
        if (ty == hipGraphNodeTypeKernel) ++nk; else if (ty == hipGraphNodeTypeMemcpy) ++nc; else if (ty == hipGraphNodeTypeMemset) ++nm; else ++no;
      }
      fprintf(stderr, "  seg %zu: lane %d stream %d, %zu nodes (%d kernels, %d copies, %d memsets, %d other; topo %d..%d)%s, waits:", s, g.lane, g.stream, g.nodes.size(), nk, nc, nm, no,
              pos[g.nodes.front()], pos[g.nodes.back()], g.record ? ", records" : "");
      for (int w : g.waits) fprintf(stderr, " %d", w);
      fprintf(stderr, "\n");
    }
  }
  *out = (miseg_graph_split_t)plan;
  return MISEG_OK;
}

extern "C" int miseg_graph_split_launch(miseg_graph_split_t plan_, miseg_stream_t stream_) {
  Plan* p = (Plan*)plan_;
  MISEG_REQUIRE(p, MISEG_E_BADARG, "graph_split_launch: null plan");
  // The main lane runs on the CALLER's stream.  A stream that sits on a wait for an event of another stream costs the streams that are
  // running 1.3 us on every kernel (scripts/debug/graph_two_queue_probe2.py: 400 small kernels 760 -> 1280 us while a bystander stream waits
  // for their end; the same with the pieces on a stream of the plan's own and the caller waiting for it - and what the runtime's own replay
  // of a multi-stream graph does to its launch stream).  On the caller's stream nothing waits for the main lane.
  hipStream_t caller = (hipStream_t)stream_;
  hipStream_t main = caller;
  auto st = [&](int s) { return s == 0 ? main : p->side[s - 1]; };
  GS_CHECK(hipEventRecord(p->start, caller));
  for (auto s : p->side) GS_CHECK(hipStreamWaitEvent(s, p->start, 0));
  for (auto& s : p->segs) {
    for (int w : s.waits) GS_CHECK(hipStreamWaitEvent(st(s.stream), p->segs[w].done, 0));
    GS_CHECK(hipGraphLaunch(s.exec, st(s.stream)));
    if (s.record) GS_CHECK(hipEventRecord(s.done, st(s.stream)));
  }
  for (size_t i = 0; i < p->side.size(); ++i) {      // the main lane's stream ends behind the side streams
    GS_CHECK(hipEventRecord(p->side_end[i], p->side[i]));
    GS_CHECK(hipStreamWaitEvent(main, p->side_end[i], 0));
  }
  GS_CHECK(hipEventRecord(p->tail, main));
  p->launched = true;
  return MISEG_OK;
}

extern "C" void miseg_graph_split_destroy(miseg_graph_split_t plan_) { destroy_plan((Plan*)plan_); }

// 1: kernels launched on the two streams were seen to run side by side (they sit on different hardware queues); 0: one after the other.
// What a device-side wait between two streams (miseg_flag_wait) needs to be told before it is used: a waiter that shares its producer's
// hardware queue blocks the kernel that would release it and runs into its timeout.
extern "C" int miseg_streams_run_concurrently(miseg_stream_t a_, miseg_stream_t b_) {
  hipStream_t a = (hipStream_t)a_, b = (hipStream_t)b_;
  MISEG_REQUIRE(a != b, MISEG_E_BADARG, "streams_run_concurrently: one stream given twice");
  hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return miseg::set_error(MISEG_E_LAUNCH, "streams_run_concurrently: hipEventCreate failed");
  }
  const bool ok = run_concurrently(a, b, e0, e1, eb);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eb);
  return ok ? 1 : 0;
}
