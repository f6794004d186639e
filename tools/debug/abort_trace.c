/* LD_PRELOAD shim (debugging aid): on SIGABRT / SIGSEGV / SIGBUS write a native backtrace to $ABORT_TRACE_FILE (default
 * /tmp/abort_trace.txt) before the default action — pytest's fd capture swallows whatever the runtime printed before abort().
 * gcc -shared -fPIC -O1 -o tools/debug/libabort_trace.so tools/debug/abort_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig) {
    const char *path = getenv("ABORT_TRACE_FILE");
    int fd = open(path ? path : "/tmp/abort_trace.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        const char *m = sig == SIGABRT ? "SIGABRT\n" : (sig == SIGSEGV ? "SIGSEGV\n" : "SIGBUS\n");
        (void)!write(fd, m, strlen(m));
        void *bt[96];
        int n = backtrace(bt, 96);
        backtrace_symbols_fd(bt, n, fd);
        /* what the runtime printed before aborting sits in pytest's capture file behind fd 2 (and fd 1): copy the tail */
        for (int src = 2; src >= 1; src--) {
            char link[32]; static char buf[8192];
            link[0] = 0; strcat(link, "/proc/self/fd/"); link[14] = (char)('0' + src); link[15] = 0;
            int in = open(link, O_RDONLY);
            if (in < 0) continue;
            off_t end = lseek(in, 0, SEEK_END);
            if (end > 0) {
                off_t from = end > (off_t)sizeof buf ? end - (off_t)sizeof buf : 0;
                lseek(in, from, SEEK_SET);
                ssize_t k = read(in, buf, sizeof buf);
                const char *hdr = src == 2 ? "\n--- tail of fd 2 ---\n" : "\n--- tail of fd 1 ---\n";
                (void)!write(fd, hdr, strlen(hdr));
                if (k > 0) (void)!write(fd, buf, (size_t)k);
            }
            close(in);
        }
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    void *warm[4]; (void)backtrace(warm, 4);          /* loads libgcc now, not inside the handler */
    struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = handler; sa.sa_flags = SA_NODEFER | SA_ONSTACK;
    sigaction(SIGABRT, &sa, 0); sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0);
}
