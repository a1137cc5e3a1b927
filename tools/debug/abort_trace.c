/* LD_PRELOAD helper (debugging only): interposes abort() and prints the calling thread's native backtrace first.
 *   gcc -shared -fPIC -O1 -g tools/debug/abort_trace.c -ldl -o tools/debug/abort_trace.so
 *   LD_PRELOAD=$PWD/tools/debug/abort_trace.so python -m pytest -p no:faulthandler ... */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void dump(const char* why) {
    void* frames[96];
    if (write(2, why, strlen(why))) {}
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
}

void abort(void) {
    dump("\n=== abort() called; native backtrace ===\n");
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
    _exit(134);
}

static void on_fatal(int sig) {
    dump(sig == SIGSEGV ? "\n=== SIGSEGV; native backtrace ===\n" : "\n=== fatal signal; native backtrace ===\n");
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_fatal;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}
