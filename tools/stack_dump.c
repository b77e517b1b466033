// tools/stack_dump.c -- LD_PRELOAD helper for the image (no gdb on PATH): when the process still runs MVFX_STACKDUMP_AFTER seconds after its start,
// every thread prints its backtrace (SIGUSR2 to each thread in turn), then the process exits with status 99.  For pipelines that hang.
//   gcc -shared -fPIC -O1 -g tools/stack_dump.c -o /tmp/libstackdump.so -lpthread
//   MVFX_STACKDUMP_AFTER=40 MVFX_GST_LD_PRELOAD=/tmp/libstackdump.so python tools/...   (tests/gst_env.py passes the preload on to gst-launch-1.0)
#define _GNU_SOURCE
#include <dirent.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static void handler(int sig)
{
    void *frames[48];
    char line[96];
    int n = snprintf(line, sizeof line, "\n== thread %ld\n", (long)syscall(SYS_gettid));
    (void)!write(2, line, (size_t)n);
    n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, 2);
    (void)sig;
}

static void *watchdog(void *arg)
{
    const int after = (int)(long)arg;
    sleep((unsigned)after);
    char line[200];
    int n = snprintf(line, sizeof line, "\n#### stack_dump: still running after %d s -- every thread's backtrace follows\n", after);
    (void)!write(2, line, (size_t)n);
    const long self = (long)syscall(SYS_gettid);
    DIR *d = opendir("/proc/self/task");
    struct dirent *e;
    while (d && (e = readdir(d))) {
        const long tid = atol(e->d_name);
        if (tid <= 0 || tid == self) continue;
        syscall(SYS_tgkill, getpid(), tid, SIGUSR2);
        usleep(150000);
    }
    if (d) closedir(d);
    FILE *maps = fopen("/proc/self/maps", "r");
    if (maps) {
        while (fgets(line, sizeof line, maps))
            if (strstr(line, "r-xp") && (strstr(line, "libmi355") || strstr(line, "libgst") || strstr(line, "libmvfx") || strstr(line, "libhsa") || strstr(line, "libamdhip")))
                (void)!write(2, line, strlen(line));
        fclose(maps);
    }
    _exit(99);
    return NULL;
}

__attribute__((constructor)) static void install(void)
{
    const char *e = getenv("MVFX_STACKDUMP_AFTER");
    if (!e || atoi(e) <= 0) return;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGUSR2, &sa, NULL);
    pthread_t t;
    pthread_create(&t, NULL, watchdog, (void *)(long)atoi(e));
    pthread_detach(t);
}
