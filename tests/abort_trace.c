/* Test infrastructure only: a SIGABRT / SIGSEGV / SIGBUS handler that writes the NATIVE backtrace of the faulting thread to stderr and
 * then hands over to the handler that was installed before it (pytest's faulthandler, which prints the Python frames).  A GPU-side
 * fault is reported by the ROCm runtime's own threads through abort(): Python's faulthandler alone shows no frame for them.
 * Built and loaded by tests/conftest.py on GPU runs (gcc is part of the image); never part of the product path. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_prev[64];
static int g_fd = -1;   /* a file of our own: pytest captures fd 2 per test and never reports it when the process dies */

static void handler(int sig, siginfo_t* info, void* ctx) {
  static const char head[] = "\n=== native backtrace (tests/abort_trace.c) ===\n";
  void* frames[64];
  static const char tail[] = "=== end of native backtrace ===\n";
  int n = backtrace(frames, 64);
  for (int pass = 0; pass < 2; ++pass) {
    const int fd = pass ? g_fd : 2;
    if (fd < 0) continue;
    (void)!write(fd, head, sizeof(head) - 1);
    backtrace_symbols_fd(frames, n, fd);
    (void)!write(fd, tail, sizeof(tail) - 1);
  }
  struct sigaction* p = &g_prev[sig & 63];
  if (p->sa_flags & SA_SIGINFO) {
    if (p->sa_sigaction) { p->sa_sigaction(sig, info, ctx); return; }
  } else if (p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN && p->sa_handler) {
    p->sa_handler(sig);
    return;
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

#include <fcntl.h>
int fa_install_abort_trace(const char* path) {
  if (path && *path) g_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
  const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS};
  void* warm[4];
  backtrace(warm, 4);   /* loads libgcc now, not inside the handler */
  for (unsigned i = 0; i < sizeof(sigs) / sizeof(sigs[0]); ++i) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    if (sigaction(sigs[i], &sa, &g_prev[sigs[i] & 63]) != 0) return 1;
  }
  return 0;
}
