// tools/segv_trace.c -- LD_PRELOAD helper for the image (no gdb on PATH): prints the faulting thread's backtrace on SIGSEGV / SIGBUS / SIGABRT.
//   gcc -shared -fPIC -O1 -g tools/segv_trace.c -o /tmp/libsegvtrace.so ; LD_PRELOAD=/tmp/libsegvtrace.so gst-launch-1.0 -f ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

static void handler(int sig, siginfo_t *info, void *uctx)
{
    void *frames[64];
    char line[160];
    const ucontext_t *uc = (const ucontext_t *)uctx;
    int n = snprintf(line, sizeof line, "\n== signal %d, fault address %p, rip %p\n", sig, info->si_addr,
                     (void *)uc->uc_mcontext.gregs[REG_RIP]);
    (void)!write(2, line, (size_t)n);
    n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    FILE *maps = fopen("/proc/self/maps", "r");
    if (maps) {
        while (fgets(line, sizeof line, maps))
            if (strstr(line, "r-xp") && (strstr(line, "libmi355") || strstr(line, "libamdhip") || strstr(line, "libgst") || strstr(line, "libmvfx")))
                (void)!write(2, line, strlen(line));
        fclose(maps);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
    sigaltstack(&ss, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}
