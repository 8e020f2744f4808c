// Command-list executor: replays a recorded sequence of this library's own C-ABI entry points in one call.
//
// Why: the 2D branch (UNetResNet34 of mopa/models/resnet34_unet.py:131-191 behind Net2DSeg.forward, mopa/models/xmuda_arch.py:49-79)
// is ~460 kernel launches per forward + backward pass, each one a ctypes call from Python: 7.4 ms of host time per pass at the
// bench shape (round 4) -- the step follows the host on the small per-GPU batches of configs[3] / [4].  The layer program is static
// per (shape, mode): mopa_amd/dense2d.py::Graph2D already records the pass once (under a stream capture, which pins every
// activation, workspace and weight-form address in a private memory pool) -- here the recording is a flat list of (entry point,
// argument slots) plus the event hand-overs between the main and the weight-gradient stream, and a replay is ONE call that walks it:
// the same entry points, the same arguments, the same order -- identical bits -- at the HIP runtime's launch cost instead of the
// interpreter's.  (Replaying the captured hipGraph itself costs MORE host time than the eager pass on this runtime: DESIGN.md
// section 5, "HIP-graph replay of the 2D backbone".)
//
// The list lives in caller memory (a host array of int64 words); no allocation, no state in the library.
//   word stream:  op, nargs, arg[0] .. arg[nargs-1],  op, nargs, ...
//   op >= 0   entry point id (mopa_exec_fn_id); integers and pointers travel as int64, float / double as the bits of a double
//   op == -1  hipEventRecord(event = arg[0], stream = arg[1])
//   op == -2  hipStreamWaitEvent(stream = arg[0], event = arg[1])
#include "common.h"
#pragma GCC visibility push(default)
#include "../../include/mopa_hip.h"
#pragma GCC visibility pop
#include <string.h>

static inline double slot_f(int64_t v) {
  double d;
  memcpy(&d, &v, sizeof(d));
  return d;
}
#include "exec_table.inc"

// id of a launching entry point by name (-1: unknown or not replayable) / how many there are
MOPA_API int mopa_exec_fn_id(const char* name_host) {
  if (!name_host) return -1;
  for (int i = 0; i < EXEC_N; ++i)
    if (strcmp(EXEC_NAMES[i], name_host) == 0) return i;
  return -1;
}
MOPA_API int mopa_exec_fn_count(void) { return EXEC_N; }

// Replay `n_words` words of commands.  Returns 0, or the first non-zero return code with *fail_word_host (optional) = the word
// index of the failing command.
MOPA_API int mopa_exec_replay(const int64_t* cmds_host, int64_t n_words, int64_t* fail_word_host) {
  if (!cmds_host || n_words < 0) return MOPA_ERR_ARG;
  int64_t i = 0;
  while (i < n_words) {
    if (i + 2 > n_words) return MOPA_ERR_ARG;
    const int64_t op = cmds_host[i], nargs = cmds_host[i + 1];
    if (nargs < 0 || nargs > 40 || i + 2 + nargs > n_words) return MOPA_ERR_ARG;
    const int64_t* a = cmds_host + i + 2;
    int rc;
    if (op == -1) rc = (nargs == 2 && hipEventRecord((hipEvent_t)a[0], (hipStream_t)a[1]) == hipSuccess) ? MOPA_OK : MOPA_ERR_LAUNCH;
    else if (op == -2) rc = (nargs == 2 && hipStreamWaitEvent((hipStream_t)a[0], (hipEvent_t)a[1], 0) == hipSuccess) ? MOPA_OK : MOPA_ERR_LAUNCH;
    else rc = exec_dispatch((int)op, a, (int)nargs);
    if (rc != MOPA_OK) {
      if (fail_word_host) *fail_word_host = i;
      return rc;
    }
    i += 2 + nargs;
  }
  return MOPA_OK;
}
