/* LD_PRELOAD helper for tools/graph_try.py: on SIGSEGV / SIGBUS / SIGABRT print the faulting thread's call stack
 * (glibc backtrace: module + offset per frame, resolvable with addr2line / nm) and the fault address to stderr, then
 * re-raise with the default action.  gcc -shared -fPIC -O1 -g tools/segv_bt.c -o tools/segv_bt.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    char buf[128];
    int here = 0;
    int n = snprintf(buf, sizeof(buf), "\nSEGV_BT signal %d fault address %p (handler stack near %p)\n", sig, si ? si->si_addr : (void*)0, (void*)&here);
    if (n > 0) (void)!write(2, buf, (size_t)n);
    /* a runaway recursion is hundreds of thousands of frames deep: what matters is both ends of the stack - the innermost
     * frames (the function that recurses) and the OUTERMOST (which API call it was entered from) */
    static void* frames[400000];
    const int depth = backtrace(frames, 400000);
    n = snprintf(buf, sizeof(buf), "SEGV_BT %d frames; innermost 6:\n", depth);
    if (n > 0) (void)!write(2, buf, (size_t)n);
    backtrace_symbols_fd(frames, depth < 6 ? depth : 6, 2);
    if (depth > 6) {
        const int tail = depth - 6 < 40 ? depth - 6 : 40;
        n = snprintf(buf, sizeof(buf), "SEGV_BT ... outermost %d:\n", tail);
        if (n > 0) (void)!write(2, buf, (size_t)n);
        backtrace_symbols_fd(frames + depth - tail, tail, 2);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    /* an alternate stack: a stack overflow (runaway recursion) leaves no room for the handler on the thread's own */
    static char alt[1 << 18];
    stack_t ss;
    ss.ss_sp = alt, ss.ss_size = sizeof(alt), ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
    /* prime backtrace(): its first call loads libgcc, which is not async-signal-safe */
    void* f[2];
    (void)backtrace(f, 2);
}
