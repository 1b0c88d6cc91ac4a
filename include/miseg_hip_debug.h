/* libmiseg_hip.so - measurement and experiment entry points (NOT part of the product ABI of include/miseg_hip.h: no product path of
 * mi-seg_amd/ calls them; bench.py's roofline leg, MISEG_STEP_STAMPS and the opt-in MISEG_GRAPH_SPLIT replay do).  They are exported by the
 * same shared object; miseg_prof_* only do something in the measurement build libmiseg_hip_prof.so (csrc/build.py links that variant with
 * -Wl,--wrap=hipLaunchKernel; the product library is linked without it and answers MISEG_E_UNSUPPORTED). */
#ifndef MISEG_HIP_DEBUG_H
#define MISEG_HIP_DEBUG_H
#include "miseg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The stream waits ON THE DEVICE (a one-thread kernel that spins with s_sleep) until *flag_dev >= *want_dev - e.g. a flag another
 * stream, or another hipGraph launch, sets with miseg_counter_copy once its producers have run: an ordering between two captured graphs that
 * needs no edge between them.  NO FORWARD-PROGRESS GUARANTEE BY ITSELF: HIP streams share a few hardware queues (4 on ROCm 7.2), and a
 * waiter that sits on its producer's queue blocks the kernel that would set the flag.  Use it only between streams for which
 * miseg_streams_run_concurrently() returned 1.  After timeout_us (<= 2 s) the kernel gives up, increments *timed_out_dev (may be null)
 * and returns: what follows then runs on data that may be incomplete - a non-zero counter is an ERROR the caller must raise, not a
 * statistic.  Experiment of round 4; nothing in the runtime uses it. */
int miseg_flag_wait(const uint64_t* flag_dev, const uint64_t* want_dev, uint64_t timeout_us, uint32_t* timed_out_dev, miseg_stream_t stream);
/* 1 when kernels launched on the two streams run side by side (different hardware queues), 0 when one after the other (a 100 us spin
 * kernel on each, timed together); negative MISEG_E_* on failure.  Synchronises both streams. */
int miseg_streams_run_concurrently(miseg_stream_t a, miseg_stream_t b);
/* measurement aid: *slot_dev = the device's constant-rate wall clock (100 MHz on gfx950) when the stream reaches this point.  A one-thread
 * kernel, so it can be recorded into a hipGraph: the order in which the streams of a replayed step reach their joins is visible without a
 * tracer (whose per-dispatch cost reorders exactly that).  Host side: MISEG_STEP_STAMPS=1|2, hip/ops.py::stamp; bench.py prints them. */
int miseg_debug_stamp(uint64_t* slot_dev, miseg_stream_t stream);

/* measurement aid (bench.py's roofline leg): while armed with a tag >= 0, EVERY kernel this library launches records its own begin / end
 * timestamps (hipExtLaunchKernel start / stop events: the dispatch's own clock, what rocprofv3 --kernel-trace reports) - once, in place, on
 * its own stream, beside whatever else runs.  miseg_prof_arm(-1) disarms.  miseg_prof_read waits for the recorded launches, writes up to
 * `max` (tag, milliseconds) pairs in launch order, forgets them and returns how many there were.  Not for use under stream capture. */
int miseg_prof_arm(int tag);
int miseg_prof_read(int* tags, float* ms, int max);

/* A captured multi-stream hipGraph replayed as single-stream graphs (csrc/graphsplit.cpp, ABI 7; opt-in, MISEG_GRAPH_SPLIT=1 in
 * runtime/graph.py).  miseg_graph_split_create takes the hipGraph_t of a finished capture (torch: CUDAGraph(keep_graph=True)
 * .raw_cuda_graph()), decomposes it into chains, cuts them at the edges that cross between chains and instantiates every piece as a graph
 * of its own; miseg_graph_split_launch replays the pieces - the longest chain on `stream` itself, the others on streams the plan owns (at
 * most max_side_streams, else MISEG_E_UNSUPPORTED; picked at creation so that they demonstrably run beside `launch_stream`, see
 * miseg_graph_split_info.streams_concurrent), one event per
 * crossing edge - in the partial order of the captured graph, and leaves `stream` waiting for all of it.  The memory pool the captured
 * nodes point into must outlive the plan's launches; the plan itself holds clones of the graph.  An alternative to the runtime's
 * hipGraphLaunch of the whole graph, which leaves the launch stream waiting on its internal streams for the whole replay (what that
 * costs, and what this buys where: csrc/graphsplit.cpp). */
typedef void* miseg_graph_split_t;
typedef struct {
  int nodes, lanes, segments, crossing_edges, side_streams, main_lane_nodes;
  int streams_concurrent;      /* 1: the plan's streams were seen to run beside the launch stream and each other (else pieces may serialise) */
} miseg_graph_split_info;
int miseg_graph_split_create(void* hip_graph, miseg_stream_t launch_stream, int max_side_streams, miseg_graph_split_t* out,
                             miseg_graph_split_info* info /* may be null */);
int miseg_graph_split_launch(miseg_graph_split_t plan, miseg_stream_t stream);
void miseg_graph_split_destroy(miseg_graph_split_t plan);

/* 1 in libmiseg_hip_prof.so, 0 in the product library */
int miseg_prof_available(void);

#ifdef __cplusplus
}
#endif
#endif
