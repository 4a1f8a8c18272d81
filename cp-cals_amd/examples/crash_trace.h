// Fatal-signal evidence for the CLI driver and the C++ test programs: on SIGSEGV / SIGBUS / SIGABRT /
// SIGFPE / SIGILL the process writes the signal and a backtrace (backtrace_symbols_fd: async-signal-
// safe, no malloc) to stderr, then re-raises with the default action so the exit status still says
// which signal it was.  A crash during process teardown (after main returned) is covered too: the
// handler stays installed until the process is gone.  Link with -rdynamic for symbol names.
#ifndef CALS_AMD_CRASH_TRACE_H
#define CALS_AMD_CRASH_TRACE_H

#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

namespace crash_trace {
inline void on_fatal(int sig) {
  const char *name = sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : sig == SIGABRT ? "SIGABRT"
                   : sig == SIGFPE ? "SIGFPE" : "fatal signal";
  const char head[] = "\n*** crash_trace: ";
  const char tail[] = " -- backtrace of the faulting thread:\n";
  ssize_t w = write(2, head, sizeof(head) - 1);
  w = write(2, name, strlen(name));
  w = write(2, tail, sizeof(tail) - 1);
  (void)w;
  void *frames[96];
  const int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
inline void install() {
  void *warm[2];
  (void)backtrace(warm, 2);  // loads libgcc's unwinder now, not inside the handler
  for (int sig : {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL}) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_fatal;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_NODEFER;
    sigaction(sig, &sa, nullptr);
  }
}
}  // namespace crash_trace
#endif
