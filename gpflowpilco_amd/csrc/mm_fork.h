// One side stream and a (fork, join) event pair per device, created on first use: independent, latency-bound kernel chains run on it
// BESIDE a long sweep of the main stream (mm_kernels.hip: the forward's moment chain beside the diagonal reduce; mm_compose_bwd.hip:
// the backward's aggregate chain beside the diagonal sweep).  The events are shared by the device's callers: every enqueue sequence
// that touches them holds `seq`.  The side stream is in order, so a waiter on `join` that arrives after a LATER record only waits
// longer, never too little.  Under stream capture the side stream joins the capture through the events (fork / join inside one
// call only: a call that cannot join before it returns does not fork while capturing).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

struct MMFork {
  hipStream_t s2;
  hipEvent_t fork, join;
  bool ok;
  std::recursive_mutex seq;     // (recursive: a sequence that holds it calls launchers that join-wait themselves)
};
MMFork* mm_fork_get();      // nullptr if the stream / events could not be created (callers then stay on the main stream)
// Make `stream` wait for everything the side stream has been given so far (the q stage's moment chain): called by whoever reads
// s12 / the moment table or overwrites them.  A no-op when nothing was ever forked.  0 or an error code.
int mm_fork_join_wait(hipStream_t stream);
