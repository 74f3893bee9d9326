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
  static thread_local bool slack_set = false;
  if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 2000UL); slack_set = true; }     // the default slack of 50 us would triple a 20 us sleep
  for (int i = 0;; i++) {
    hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (i < 8) { sched_yield(); continue; }
    timespec ts{0, i < 64 ? 20000 : 100000};      // 20 us, then 100 us (plus the kernel's timer slack)
    nanosleep(&ts, nullptr);
  }
}
}  // namespace zkhip
