// Waiting for a HIP event without occupying a host core.
// hipEventSynchronize - also on events created with hipEventBlockingSync - was measured to keep the waiting thread busy on the
// GPU box (six prover threads of the streaming prover: six cores at 93 %, zk-prover threads in tools' per-thread accounting),
// and a one-GPU job gets sixteen cores.  The waits here are long next to a timer tick (a proof's MSM phase is milliseconds, a
// witness launch tens of milliseconds), so: poll the event, a few yields first (short waits return at once), then sleep.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>

namespace zkhip {
inline hipError_t zk_event_wait(hipEvent_t ev) {
  hipError_t e = hipEventQuery(ev);
  if (e != hipErrorNotReady) return e;
  // The sleeps below want a timer slack of ~2 us (the default 50 us would triple a 20 us sleep).  The slack is a property of the
  // calling THREAD, which may be the application's: it is changed for the duration of this wait only and put back afterwards.
  const int old_slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
  if (old_slack > 2000) (void)prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0);
  for (int i = 0;; i++) {
    if (i < 8) sched_yield();
    else {
      timespec ts{0, i < 64 ? 20000 : 100000};      // 20 us, then 100 us (plus the timer slack)
      nanosleep(&ts, nullptr);
    }
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) break;
  }
  if (old_slack > 2000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old_slack, 0, 0, 0);
  // hipErrorNotReady is a status, not a failure, but the runtime may have recorded it as this thread's last error: a later
  // hipGetLastError() check of a kernel launch must not trip over it
  if (e == hipSuccess) (void)hipGetLastError();
  return e;
}
}  // namespace zkhip
