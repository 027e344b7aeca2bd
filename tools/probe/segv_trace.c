/* segv_trace.c -- a native backtrace when a test process dies of SIGSEGV / SIGABRT / SIGBUS (Python's faulthandler prints the Python
 * frames only).  Soak tooling: tests/conftest.py loads it when MDVT_SEGV_TRACE=1 (built on the fly with gcc).  The handler prints
 * the faulting thread's frames as "<mapped file>(+offset)" (resolve with `addr2line -f -e <file> <offset>`), then hands the
 * signal on to whoever held it before (faulthandler).
 *   gcc -O1 -g -shared -fPIC -o /tmp/libsegv_trace.so tools/probe/segv_trace.c                                                */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction old_segv, old_abrt, old_bus;
static int out_fd = 2;          /* pytest captures fd 2: the trace goes to the file given at install time */

static void on_signal(int sig, siginfo_t* info, void* uc)
{
    static const char head[] = "\n== native backtrace of the faulting thread (segv_trace) ==\n";
    void* frames[64];
    (void)!write(out_fd, head, sizeof head - 1);
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, out_fd);
    struct sigaction* old = sig == SIGSEGV ? &old_segv : (sig == SIGABRT ? &old_abrt : &old_bus);
    if (old->sa_flags & SA_SIGINFO) { if (old->sa_sigaction) { old->sa_sigaction(sig, info, uc); return; } }
    else if (old->sa_handler != SIG_DFL && old->sa_handler != SIG_IGN) { old->sa_handler(sig); return; }
    signal(sig, SIG_DFL);
    raise(sig);
}

void segv_trace_install(const char* path)
{
    if (path && path[0]) { const int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) out_fd = fd; }
    void* warm[2];
    backtrace(warm, 2);                         /* (loads libgcc now, not inside the handler) */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_signal;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGABRT, &sa, &old_abrt);
    sigaction(SIGBUS, &sa, &old_bus);
}
