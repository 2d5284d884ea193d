// A side stream and a (fork, join) event pair PER CALLER STREAM, created on first use: independent, latency-bound kernel chains
// run on it BESIDE a long sweep of the caller's stream (mm_kernels.hip: the forward's moment chain beside the diagonal reduce;
// mm_compose_bwd.hip: the backward's aggregate chain beside the diagonal sweep).  Round 4 kept ONE triple per device, shared by
// every caller stream under a process-wide mutex: two streams of one process serialised on the mutex, picked up each other's joins
// as false dependencies, and a stream under capture could be made to wait on an event another stream recorded outside the
// capture.  Now the state belongs to the caller's stream: `mm_fork_get(stream)` looks the stream's triple up in a small table
// (keyed by device and stream handle; the table's lock covers the lookup only, never an enqueue), so calls on different streams
// share nothing -- one of them may be under HIP-graph capture (its side stream joins THAT capture through its own events) while
// another runs eagerly.  The contract per stream is the stream-ordered one: one enqueuing thread per stream at a time.  Every call
// that forks joins before it returns, so nothing is ever left on a side stream across calls.
#pragma once
#include <hip/hip_runtime.h>

struct MMFork {
  hipStream_t s2;
  hipEvent_t fork, join;
};
// The caller stream's fork state; created when absent (nullptr if that failed: callers then stay on their own stream).
MMFork* mm_fork_get(hipStream_t stream);
// Make `stream` wait for what ITS side stream has been given so far (the q stage's moment chain): called by whoever reads s12 /
// the moment table or overwrites them inside a call that may have forked.  A no-op for a stream that never forked (and for a side
// stream itself: it is in order).  0 or a hipError_t.
int mm_fork_join_wait(hipStream_t stream);
