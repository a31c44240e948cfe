/* Debug aid (round 4): a SIGSEGV handler that writes the NATIVE backtrace of the faulting thread (glibc backtrace_symbols_fd) and the
 * fault address to a FILE (pytest redirects fd 2), running on an alternate stack so that it also fires on stack exhaustion, then
 * re-raises with the default action. Loaded when LLMREC_SEGV_BT=<file>. Not part of the product. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static int g_fd = 2;
static void handler(int sig, siginfo_t* info, void* ctx) {
    (void)ctx;
    char buf[160];
    int n = snprintf(buf, sizeof buf, "\n[segv_bt] signal %d, fault address %p, stack var at %p, native backtrace:\n", sig, info ? info->si_addr : (void*)0, (void*)buf);
    if (n > 0) { ssize_t w = write(g_fd, buf, (size_t)n); (void)w; }
    void* frames[96];
    int k = backtrace(frames, 96);
    backtrace_symbols_fd(frames, k, g_fd);
    fsync(g_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}
int segv_bt_install(const char* path) {
    static char* alt = 0;
    if (path && *path) { int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) g_fd = fd; }
    if (!alt) {
        alt = (char*)malloc(1 << 18);
        stack_t ss; ss.ss_sp = alt; ss.ss_size = 1 << 18; ss.ss_flags = 0;
        sigaltstack(&ss, 0);
    }
    void* warm[4]; backtrace(warm, 4);                    /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGSEGV, &sa, 0);
}
