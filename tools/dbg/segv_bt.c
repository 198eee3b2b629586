/* LD_PRELOAD helper (debugging only): print a native backtrace on SIGSEGV / SIGABRT / SIGBUS.
 *   gcc -shared -fPIC -o /tmp/segv_bt.so tools/dbg/segv_bt.c
 *   LD_PRELOAD=/tmp/segv_bt.so python -m pytest -p no:faulthandler ...        (pytest's faulthandler would replace the handler) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static int out_fd = 2;     /* a dup of the ORIGINAL stderr: pytest's fd-level capture replaces fd 2 during tests */

static void handler(int sig, siginfo_t* si, void* ctx) {
    void* frames[96];
    char msg[128];
    int n = snprintf(msg, sizeof msg, "\n==== signal %d at address %p: native backtrace ====\n", sig, si ? si->si_addr : 0);
    if (write(out_fd, msg, n) < 0) {}
    n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, out_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    out_fd = dup(2);
    void* warm[4];
    backtrace(warm, 4);          /* loads libgcc now, not inside the handler */
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
    sigaltstack(&ss, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
