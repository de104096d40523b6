/* LD_PRELOAD helper for debugging: print a native backtrace when the process receives SIGABRT / SIGSEGV.
 *   gcc -shared -fPIC -o abort_bt.so abort_bt.c && LD_PRELOAD=./abort_bt.so python ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>

static void handler(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    dprintf(2, "\n[abort_bt] signal %d, native backtrace (%d frames):\n", sig, n);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void) {
    signal(SIGABRT, handler);
    signal(SIGSEGV, handler);
}
