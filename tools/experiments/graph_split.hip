// hamt_graph_split: run a captured hipGraph's parallel branches on SEPARATE streams.
//
// Why (measured on MI355X / ROCm 7.0, tools/graph_branch_probe.py): a captured graph with two independent chains of 100 small
// kernels each (fork / join inside the capture) replays in 0.53 ms -- one chain alone takes 0.18 ms, the two chains as two graphs
// on two streams 0.19 ms.  Inside ONE graph the runtime neither overlaps parallel branches nor keeps the single-chain launch rate,
// whatever the capture order and whatever DEBUG_HIP_FORCE_GRAPH_QUEUES says; in the training step the panorama encoder's backward
// started 1.4 ms after its inputs were ready, behind all nine text layers' backward (profiles/r04_queues_*.txt).
//
// What this does: the captured graph (hipGraph_t, e.g. torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()) is read back
// (nodes, edges, node parameters), transitively reduced, cut into maximal LINEAR chains ("segments": a node continues its
// predecessor's segment iff it is that node's only successor and has no other predecessor), every segment becomes a small
// graph of its own (kernel / memcpy / memset / empty nodes re-added in chain order) and is assigned to one of `n_streams`
// streams (a segment continues on the stream of a predecessor whose stream no sibling has taken, else takes a free one).
// hamt_graph_split_launch then launches the segment graphs in topological order on their streams, with hipStreamWaitEvent /
// hipEventRecord for exactly the dependencies that cross streams; streams[0] is the caller's stream: every other stream
// first waits for it and it finally waits for all of them, so the call is stream-ordered like hipGraphLaunch.
//
// Host-only code (the HIP graph API); the original graph must stay alive as long as the split (kernel parameters of the
// re-added nodes are copied by hipGraphAddKernelNode, but `extra`-style launches keep pointing into the original node).
#include "common.h"
#include <algorithm>
#include <map>
#include <string>
#include <string.h>
#include <vector>

namespace {

struct Segment {
  std::vector<int> nodes;            // original node indices, chain order
  std::vector<int> deps;             // segments that must have finished (transitively reduced)
  int stream = 0;
  bool record = false;               // some dependent runs on another stream
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipEvent_t ev = nullptr;
  uint32_t* sig = nullptr;           // 8 bytes of signal memory: the launch epoch once the segment has run (stream memory operations)
};

}  // namespace

struct hamt_graph_exec {
  std::vector<Segment> segs;         // launch order = topological order
  int n_streams = 1, n_nodes = 0, n_edges = 0, n_cross = 0;
  std::vector<int> stream_nodes;
  std::vector<char> stream_used;
  hipEvent_t ev_start = nullptr;
  std::vector<hipEvent_t> ev_end;
  std::vector<hipGraphNode_t> nodes;   // the captured graph's nodes (describe)
  std::vector<int> order;              // launch order of the segments (a topological order)
  bool use_values = true;              // cross-stream dependencies by hipStreamWriteValue32 / hipStreamWaitValue32 (else events)
  uint32_t epoch = 0;
  uint32_t* sig_start = nullptr;
  std::vector<uint32_t*> sig_end;
};

#define HAMT_HIP_OK(call, what)                                                                      \
  do {                                                                                               \
    hipError_t e__ = (call);                                                                         \
    if (e__ != hipSuccess) {                                                                         \
      hamt_set_error("hamt_graph_split: %s failed: %s", what, hipGetErrorString(e__));               \
      return HAMT_ERR_LAUNCH;                                                                        \
    }                                                                                                \
  } while (0)

// A segment's graph as a CLONE of the captured graph with every other node destroyed: exact copies of the nodes whatever their
// kind (a captured hipMemcpyAsync is a 1-D memcpy node whose parameters hipGraphMemcpyNodeGetParams does not return in a form
// hipGraphAddMemcpyNode accepts on ROCm 7.0).  O(nodes) API calls per segment, at capture time only.
static int clone_segment_graph(Segment& sg, hipGraph_t g, const std::vector<hipGraphNode_t>& nodes, const std::vector<int>& seg_of, int s) {
  HAMT_HIP_OK(hipGraphClone(&sg.graph, g), "hipGraphClone");
  for (size_t v = 0; v < nodes.size(); ++v) {
    if (seg_of[v] == s) continue;
    hipGraphNode_t cn = nullptr;
    HAMT_HIP_OK(hipGraphNodeFindInClone(&cn, nodes[v], sg.graph), "hipGraphNodeFindInClone");
    HAMT_HIP_OK(hipGraphDestroyNode(cn), "hipGraphDestroyNode");
  }
  size_t left = 0;
  HAMT_HIP_OK(hipGraphGetNodes(sg.graph, nullptr, &left), "hipGraphGetNodes");
  if (left != sg.nodes.size()) {
    hamt_set_error("hamt_graph_split: a segment's clone kept %zu nodes instead of %zu", left, sg.nodes.size());
    return HAMT_ERR_LAUNCH;
  }
  HAMT_HIP_OK(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0), "hipGraphInstantiate");
  return HAMT_OK;
}

static int build_segment_graph(Segment& sg, const std::vector<hipGraphNode_t>& nodes) {
  HAMT_HIP_OK(hipGraphCreate(&sg.graph, 0), "hipGraphCreate");
  hipGraphNode_t prev = nullptr;
  for (int v : sg.nodes) {
    hipGraphNodeType ty;
    HAMT_HIP_OK(hipGraphNodeGetType(nodes[v], &ty), "hipGraphNodeGetType");
    hipGraphNode_t nn = nullptr;
    const hipGraphNode_t* dep = prev ? &prev : nullptr;
    const size_t nd = prev ? 1 : 0;
    switch (ty) {
      case hipGraphNodeTypeKernel: {
        hipKernelNodeParams p;
        HAMT_HIP_OK(hipGraphKernelNodeGetParams(nodes[v], &p), "hipGraphKernelNodeGetParams");
        HAMT_HIP_OK(hipGraphAddKernelNode(&nn, sg.graph, dep, nd, &p), "hipGraphAddKernelNode");
        (void)hipGraphKernelNodeCopyAttributes(nodes[v], nn);      // (priority / cooperative: best effort)
        break;
      }
      case hipGraphNodeTypeMemcpy: {
        hipMemcpy3DParms p;
        HAMT_HIP_OK(hipGraphMemcpyNodeGetParams(nodes[v], &p), "hipGraphMemcpyNodeGetParams");
        if (!p.srcArray && !p.dstArray && p.extent.height <= 1 && p.extent.depth <= 1) {
          // a linear copy (what hipMemcpyAsync captures as): the 3-D form of its own parameters is rejected on re-adding
          // ("invalid pitch argument", ROCm 7.0), the 1-D node takes them
          HAMT_HIP_OK(hipGraphAddMemcpyNode1D(&nn, sg.graph, dep, nd, (char*)p.dstPtr.ptr + p.dstPos.x, (const char*)p.srcPtr.ptr + p.srcPos.x,
                                              p.extent.width, p.kind), "hipGraphAddMemcpyNode1D");
        } else {
          HAMT_HIP_OK(hipGraphAddMemcpyNode(&nn, sg.graph, dep, nd, &p), "hipGraphAddMemcpyNode");
        }
        break;
      }
      case hipGraphNodeTypeMemset: {
        hipMemsetParams p;
        HAMT_HIP_OK(hipGraphMemsetNodeGetParams(nodes[v], &p), "hipGraphMemsetNodeGetParams");
        HAMT_HIP_OK(hipGraphAddMemsetNode(&nn, sg.graph, dep, nd, &p), "hipGraphAddMemsetNode");
        break;
      }
      case hipGraphNodeTypeEmpty:
        HAMT_HIP_OK(hipGraphAddEmptyNode(&nn, sg.graph, dep, nd), "hipGraphAddEmptyNode");
        break;
      default:
        hamt_set_error("hamt_graph_split: node %d has type %d (only kernel / memcpy / memset / empty nodes are supported)", v, (int)ty);
        return HAMT_ERR_UNSUPPORTED;
    }
    prev = nn;
  }
  HAMT_HIP_OK(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0), "hipGraphInstantiate");
  return HAMT_OK;
}

extern "C" int hamt_graph_split_destroy(hamt_graph_exec* x) {
  if (!x) return HAMT_OK;
  for (auto& s : x->segs) {
    if (s.exec) hipGraphExecDestroy(s.exec);
    if (s.graph) hipGraphDestroy(s.graph);
    if (s.ev) hipEventDestroy(s.ev);
    if (s.sig) hipFree(s.sig);
  }
  if (x->sig_start) hipFree(x->sig_start);
  for (auto p : x->sig_end) if (p) hipFree(p);
  if (x->ev_start) hipEventDestroy(x->ev_start);
  for (auto e : x->ev_end) if (e) hipEventDestroy(e);
  delete x;
  return HAMT_OK;
}

extern "C" int hamt_graph_split(void* hip_graph, int n_streams, hamt_graph_exec** out) {
  HAMT_CHECK_ARG(hip_graph && out && n_streams >= 1 && n_streams <= 8, "hamt_graph_split: bad argument");
  *out = nullptr;
  hipGraph_t g = (hipGraph_t)hip_graph;
  size_t nn = 0, ne = 0;
  HAMT_HIP_OK(hipGraphGetNodes(g, nullptr, &nn), "hipGraphGetNodes");
  HAMT_CHECK_ARG(nn > 0, "hamt_graph_split: the graph has no nodes");
  std::vector<hipGraphNode_t> nodes(nn);
  HAMT_HIP_OK(hipGraphGetNodes(g, nodes.data(), &nn), "hipGraphGetNodes");
  HAMT_HIP_OK(hipGraphGetEdges(g, nullptr, nullptr, &ne), "hipGraphGetEdges");
  std::vector<hipGraphNode_t> ef(ne), et(ne);
  if (ne) HAMT_HIP_OK(hipGraphGetEdges(g, ef.data(), et.data(), &ne), "hipGraphGetEdges");
  const int N = (int)nn;
  std::map<hipGraphNode_t, int> idx;
  for (int i = 0; i < N; ++i) idx[nodes[i]] = i;
  std::vector<std::vector<int>> pred(N), succ(N);
  for (size_t e = 0; e < ne; ++e) {
    auto a = idx.find(ef[e]), b = idx.find(et[e]);
    HAMT_CHECK_ARG(a != idx.end() && b != idx.end(), "hamt_graph_split: an edge names a node outside the graph");
    succ[a->second].push_back(b->second);
    pred[b->second].push_back(a->second);
  }
  // topological order; ties by node index (= creation order of the capture), so the launch order follows the capture's
  std::vector<int> indeg(N), topo, pos(N);
  for (int i = 0; i < N; ++i) indeg[i] = (int)pred[i].size();
  {
    std::vector<int> ready;
    for (int i = 0; i < N; ++i) if (!indeg[i]) ready.push_back(i);
    auto cmp = [](int a, int b) { return a > b; };
    std::make_heap(ready.begin(), ready.end(), cmp);
    while (!ready.empty()) {
      std::pop_heap(ready.begin(), ready.end(), cmp);
      const int v = ready.back();
      ready.pop_back();
      pos[v] = (int)topo.size();
      topo.push_back(v);
      for (int s : succ[v]) if (--indeg[s] == 0) { ready.push_back(s); std::push_heap(ready.begin(), ready.end(), cmp); }
    }
  }
  HAMT_CHECK_ARG((int)topo.size() == N, "hamt_graph_split: the graph has a cycle");
  // transitive reduction (bit sets of ancestors in topological positions): an edge u -> v is redundant when u is an ancestor
  // of another predecessor of v; redundant edges would cut chains (a node with two predecessors starts a segment)
  const int W = (N + 63) / 64;
  std::vector<uint64_t> anc((size_t)N * W, 0);      // anc[v] = strict ancestors of v
  auto bit = [&](int v, int u) -> bool { return (anc[(size_t)v * W + (u >> 6)] >> (u & 63)) & 1; };
  for (int v : topo) {
    uint64_t* av = &anc[(size_t)v * W];
    for (int p : pred[v]) {
      const uint64_t* ap = &anc[(size_t)p * W];
      for (int w = 0; w < W; ++w) av[w] |= ap[w];
      av[p >> 6] |= 1ull << (p & 63);
    }
  }
  for (int v = 0; v < N; ++v) {
    std::vector<int> keep;
    for (int u : pred[v]) {
      bool redundant = false;
      for (int w : pred[v]) if (w != u && bit(w, u)) { redundant = true; break; }
      if (!redundant) keep.push_back(u);
    }
    std::sort(keep.begin(), keep.end());
    keep.erase(std::unique(keep.begin(), keep.end()), keep.end());
    pred[v] = keep;
  }
  for (int v = 0; v < N; ++v) succ[v].clear();
  for (int v = 0; v < N; ++v) for (int u : pred[v]) succ[u].push_back(v);
  // maximal linear chains
  std::vector<int> seg_of(N, -1);
  auto* x = new hamt_graph_exec();
  x->n_streams = n_streams;
  x->n_nodes = N;
  x->nodes = nodes;
  for (int v : topo) {
    if (pred[v].size() == 1 && succ[pred[v][0]].size() == 1) {
      seg_of[v] = seg_of[pred[v][0]];
      x->segs[seg_of[v]].nodes.push_back(v);
    } else {
      seg_of[v] = (int)x->segs.size();
      x->segs.emplace_back();
      x->segs.back().nodes.push_back(v);
    }
  }
  const int S = (int)x->segs.size();
  std::vector<std::vector<int>> ssucc(S);
  for (int v = 0; v < N; ++v) {
    x->n_edges += (int)pred[v].size();
    for (int u : pred[v]) {
      const int a = seg_of[u], b = seg_of[v];
      if (a != b && std::find(x->segs[b].deps.begin(), x->segs[b].deps.end(), a) == x->segs[b].deps.end()) {
        x->segs[b].deps.push_back(a);
        ssucc[a].push_back(b);
      }
    }
  }
  // streams: segments are in topological order already (created along `topo`).  A segment continues on the stream of a
  // predecessor that is the LAST segment placed on its stream and has not handed that stream to a sibling yet (the lowest such
  // stream: joins return to stream 0); else -- a fork's second child, a source -- it takes the least recently used stream.
  std::vector<int> claimed(S, 0);
  std::vector<int> last_on(n_streams, -1);
  for (int s = 0; s < S; ++s) {
    Segment& sg = x->segs[s];
    int st = -1, from = -1;
    for (int d : sg.deps) {
      const int k = x->segs[d].stream;
      if (!claimed[d] && last_on[k] == d && (st < 0 || k < st)) { st = k; from = d; }
    }
    if (st >= 0) claimed[from] = 1;
    else {
      st = 0;
      for (int k = 1; k < n_streams; ++k) if (last_on[k] < last_on[st]) st = k;
    }
    sg.stream = st;
    last_on[st] = s;
  }
  // launch order: a topological order of the segments.  Default: creation order.  HAMT_GRAPH_SPLIT_SIDE_FIRST=1: among the segments
  // whose dependencies have been launched, those on the branch streams go first (their launch then never queues behind a long
  // stream-0 segment that became ready at the same time); same-stream order is kept.
  {
    static const bool side_first = getenv("HAMT_GRAPH_SPLIT_SIDE_FIRST") != nullptr;
    std::vector<char> done(S, 0);
    for (int n_done = 0; n_done < S; ++n_done) {
      int pick = -1;
      for (int s = 0; s < S; ++s) {
        if (done[s]) continue;
        bool ready = true;
        for (int d : x->segs[s].deps) if (!done[d]) { ready = false; break; }
        // same-stream FIFO: an earlier, not yet launched segment of the same stream goes first
        for (int e = 0; e < s && ready; ++e) if (!done[e] && x->segs[e].stream == x->segs[s].stream) ready = false;
        if (!ready) continue;
        if (pick < 0) pick = s;
        else if (side_first && x->segs[pick].stream == 0 && x->segs[s].stream != 0) pick = s;
        if (!side_first) break;
      }
      done[pick] = 1;
      x->order.push_back(pick);
    }
  }
  x->stream_nodes.assign(n_streams, 0);
  x->stream_used.assign(n_streams, 0);
  for (int s = 0; s < S; ++s) {
    Segment& sg = x->segs[s];
    x->stream_nodes[sg.stream] += (int)sg.nodes.size();
    x->stream_used[sg.stream] = 1;
    for (int d : sg.deps)
      if (x->segs[d].stream != sg.stream) { x->segs[d].record = true; ++x->n_cross; }
  }
  static const bool readd = getenv("HAMT_GRAPH_SPLIT_READD") != nullptr;      // (re-add nodes from their parameters instead of cloning: kernel / memset / empty nodes only)
  for (int s = 0; s < S; ++s) {
    Segment& sg = x->segs[s];
    const int rc = readd ? build_segment_graph(sg, nodes) : clone_segment_graph(sg, g, nodes, seg_of, s);
    if (rc != HAMT_OK) { hamt_graph_split_destroy(x); return rc; }
    if (sg.record && hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming) != hipSuccess) {
      hamt_graph_split_destroy(x);
      hamt_set_error("hamt_graph_split: hipEventCreate failed");
      return HAMT_ERR_LAUNCH;
    }
  }
  // Cross-stream dependencies as STREAM MEMORY OPERATIONS on signal memory: measured (tools/fork_probe.py: a prefix on s0, then A on
  // s0 and B on s1, both ready at once, everything issued before the GPU gets there) an event wait that is still pending when it is
  // issued lets B start only when A is half done (1.26 ms against 0.97 ideal / 1.55 serialized) -- whichever of the two was issued
  // first runs first -- while hipStreamWriteValue32 behind the prefix + hipStreamWaitValue32 in front of B gives 1.05.
  {
    int can = 0;
    int devid = 0;
    (void)hipGetDevice(&devid);
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, devid);
    x->use_values = can != 0 && getenv("HAMT_GRAPH_SPLIT_EVENTS") == nullptr;
    if (x->use_values) {
      auto alloc = [&](uint32_t** p) -> bool {
        if (hipExtMallocWithFlags((void**)p, 8, hipMallocSignalMemory) != hipSuccess) return false;
        return hipMemset(*p, 0, 8) == hipSuccess;
      };
      bool ok = alloc(&x->sig_start);
      x->sig_end.assign(n_streams, nullptr);
      for (int k = 1; k < n_streams && ok; ++k) if (x->stream_used[k]) ok = alloc(&x->sig_end[k]);
      for (auto& sg : x->segs) if (sg.record && ok) ok = alloc(&sg.sig);
      if (!ok || hipDeviceSynchronize() != hipSuccess) { hamt_graph_split_destroy(x); hamt_set_error("hamt_graph_split: signal memory allocation failed"); return HAMT_ERR_LAUNCH; }
    }
  }
  if (hipEventCreateWithFlags(&x->ev_start, hipEventDisableTiming) != hipSuccess) { hamt_graph_split_destroy(x); hamt_set_error("hamt_graph_split: hipEventCreate failed"); return HAMT_ERR_LAUNCH; }
  x->ev_end.assign(n_streams, nullptr);
  for (int k = 1; k < n_streams; ++k)
    if (x->stream_used[k] && hipEventCreateWithFlags(&x->ev_end[k], hipEventDisableTiming) != hipSuccess) { hamt_graph_split_destroy(x); hamt_set_error("hamt_graph_split: hipEventCreate failed"); return HAMT_ERR_LAUNCH; }
  *out = x;
  return HAMT_OK;
}

extern "C" int hamt_graph_split_info(const hamt_graph_exec* x, int* n_nodes, int* n_segments, int* n_cross, int* stream_nodes, int n_streams) {
  HAMT_CHECK_ARG(x, "hamt_graph_split_info: null handle");
  if (n_nodes) *n_nodes = x->n_nodes;
  if (n_segments) *n_segments = (int)x->segs.size();
  if (n_cross) *n_cross = x->n_cross;
  if (stream_nodes) for (int k = 0; k < n_streams; ++k) stream_nodes[k] = k < x->n_streams ? x->stream_nodes[k] : 0;
  return HAMT_OK;
}

// per segment: {stream, number of nodes, bit mask of up to 30 dependencies RELATIVE to the segment (bit k: segment s - 1 - k)};
// returns the number of segments (<= cap are written)
extern "C" int hamt_graph_split_segments(const hamt_graph_exec* x, int* triples, int cap) {
  if (!x) return 0;
  for (int s = 0; s < (int)x->segs.size() && s < cap; ++s) {
    int m = 0;
    for (int d : x->segs[s].deps) if (s - 1 - d < 30) m |= 1 << (s - 1 - d);
    triples[3 * s] = x->segs[s].stream; triples[3 * s + 1] = (int)x->segs[s].nodes.size(); triples[3 * s + 2] = m;
  }
  return (int)x->segs.size();
}

extern "C" int hamt_graph_split_launch(hamt_graph_exec* x, void* const* streams, int n_streams) {
  HAMT_CHECK_ARG(x && streams && n_streams >= x->n_streams, "hamt_graph_split_launch: bad argument");
  for (int k = 1; k < x->n_streams; ++k)
    for (int j = 0; j < k; ++j) HAMT_CHECK_ARG(!x->stream_used[k] || streams[k] != streams[j], "hamt_graph_split_launch: streams %d and %d are the same stream", j, k);
  hipStream_t s0 = (hipStream_t)streams[0];
  bool any = false;
  for (int k = 1; k < x->n_streams; ++k) any = any || x->stream_used[k];
  if (x->use_values) {
    const uint32_t ep = ++x->epoch;
    if (any) {
      HAMT_HIP_OK(hipStreamWriteValue32(s0, x->sig_start, ep, 0), "hipStreamWriteValue32");
      for (int k = 1; k < x->n_streams; ++k)
        if (x->stream_used[k]) HAMT_HIP_OK(hipStreamWaitValue32((hipStream_t)streams[k], x->sig_start, ep, hipStreamWaitValueGte, 0xFFFFFFFFu), "hipStreamWaitValue32");
    }
    for (int s : x->order) {
      Segment& sg = x->segs[s];
      hipStream_t st = (hipStream_t)streams[sg.stream];
      for (int d : sg.deps)
        if (x->segs[d].stream != sg.stream) HAMT_HIP_OK(hipStreamWaitValue32(st, x->segs[d].sig, ep, hipStreamWaitValueGte, 0xFFFFFFFFu), "hipStreamWaitValue32");
      HAMT_HIP_OK(hipGraphLaunch(sg.exec, st), "hipGraphLaunch");
      if (sg.record) HAMT_HIP_OK(hipStreamWriteValue32(st, sg.sig, ep, 0), "hipStreamWriteValue32");
    }
    for (int k = 1; k < x->n_streams; ++k)
      if (x->stream_used[k]) {
        HAMT_HIP_OK(hipStreamWriteValue32((hipStream_t)streams[k], x->sig_end[k], ep, 0), "hipStreamWriteValue32");
        HAMT_HIP_OK(hipStreamWaitValue32(s0, x->sig_end[k], ep, hipStreamWaitValueGte, 0xFFFFFFFFu), "hipStreamWaitValue32");
      }
    return HAMT_OK;
  }
  if (any) {
    HAMT_HIP_OK(hipEventRecord(x->ev_start, s0), "hipEventRecord");
    for (int k = 1; k < x->n_streams; ++k)
      if (x->stream_used[k]) HAMT_HIP_OK(hipStreamWaitEvent((hipStream_t)streams[k], x->ev_start, 0), "hipStreamWaitEvent");
  }
  for (int s : x->order) {
    Segment& sg = x->segs[s];
    hipStream_t st = (hipStream_t)streams[sg.stream];
    for (int d : sg.deps)
      if (x->segs[d].stream != sg.stream) HAMT_HIP_OK(hipStreamWaitEvent(st, x->segs[d].ev, 0), "hipStreamWaitEvent");
    HAMT_HIP_OK(hipGraphLaunch(sg.exec, st), "hipGraphLaunch");
    if (sg.record) HAMT_HIP_OK(hipEventRecord(sg.ev, st), "hipEventRecord");
  }
  for (int k = 1; k < x->n_streams; ++k)
    if (x->stream_used[k]) {
      HAMT_HIP_OK(hipEventRecord(x->ev_end[k], (hipStream_t)streams[k]), "hipEventRecord");
      HAMT_HIP_OK(hipStreamWaitEvent(s0, x->ev_end[k], 0), "hipStreamWaitEvent");
    }
  return HAMT_OK;
}

// text description of segment `seg`: one token per node -- kernel name (up to 40 characters), M<bytes> for a memcpy, S for a
// memset, E for an empty node -- separated by ';' (debugging / profiles); returns the length written
extern "C" int hamt_graph_split_describe(const hamt_graph_exec* x, int seg, char* buf, size_t n) {
  if (!x || seg < 0 || seg >= (int)x->segs.size() || !buf || n == 0) return 0;
  std::string out;
  for (int v : x->segs[seg].nodes) {
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(x->nodes[v], &ty) != hipSuccess) { out += "?;"; continue; }
    if (ty == hipGraphNodeTypeKernel) {
      hipKernelNodeParams p;
      const char* nm = nullptr;
      if (hipGraphKernelNodeGetParams(x->nodes[v], &p) == hipSuccess && p.func) nm = hipKernelNameRefByPtr(p.func, nullptr);
      std::string t = nm ? nm : "kernel";
      if (t.size() > 60) t.resize(60);
      out += t + ";";
    } else if (ty == hipGraphNodeTypeMemcpy) {
      hipMemcpy3DParms p;
      if (hipGraphMemcpyNodeGetParams(x->nodes[v], &p) == hipSuccess) out += "M" + std::to_string(p.extent.width * (p.extent.height ? p.extent.height : 1)) + ";";
      else out += "M;";
    } else if (ty == hipGraphNodeTypeMemset) out += "S;";
    else if (ty == hipGraphNodeTypeEmpty) out += "E;";
    else out += "T" + std::to_string((int)ty) + ";";
  }
  const size_t len = std::min(out.size(), n - 1);
  memcpy(buf, out.data(), len);
  buf[len] = 0;
  return (int)len;
}
