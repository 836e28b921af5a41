/* Test infrastructure: what aborted the test process?
 *
 * A process that dies in abort() -- glibc's heap checks, an uncaught C++ exception, the GPU runtime's fault handler --
 * leaves pytest no chance to print the stderr it has captured, and Python's faulthandler shows Python frames only.
 * This handler, installed by tests/conftest.py on top of faulthandler's, appends to a file (a) the C backtrace of the
 * aborting thread and (b) the tail of whatever file descriptor 2 currently points at (pytest's capture file holds the
 * runtime's own message there), then hands the signal on.  Nothing here is part of the product.
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

static int g_fd = -1;
static int g_fd2 = -1;          /* the terminal's stderr as it was before pytest's capture, if known */
static struct sigaction g_prev;

static void put(int fd, const char *s) { if (fd >= 0) { ssize_t r = write(fd, s, strlen(s)); (void)r; } }

static void on_abort(int sig, siginfo_t *si, void *uc) {
    void *frames[96];
    const int n = backtrace(frames, 96);
    put(g_fd, "\n==== SIGABRT: C backtrace of the aborting thread ====\n");
    if (g_fd >= 0) backtrace_symbols_fd(frames, n, g_fd);
    /* the tail of fd 2, if it is a regular file that can be read back (pytest's fd-level capture) */
    struct stat st;
    if (g_fd >= 0 && fstat(2, &st) == 0 && S_ISREG(st.st_mode)) {
        static char buf[16384];
        const off_t end = lseek(2, 0, SEEK_CUR);
        if (end > 0) {
            const off_t from = end > (off_t)sizeof buf ? end - (off_t)sizeof buf : 0;
            const ssize_t got = pread(2, buf, (size_t)(end - from), from);
            put(g_fd, "==== tail of the captured stderr ====\n");
            if (got > 0) { ssize_t r = write(g_fd, buf, (size_t)got); (void)r; }
            put(g_fd, "\n==== end ====\n");
        }
    }
    if (g_fd >= 0) fsync(g_fd);
    if (g_fd2 >= 0) {                          /* the same two things where the run's log will show them */
        put(g_fd2, "\n==== SIGABRT: C backtrace of the aborting thread ====\n");
        backtrace_symbols_fd(frames, n, g_fd2);
        if (fstat(2, &st) == 0 && S_ISREG(st.st_mode)) {
            static char buf2[4096];
            const off_t end = lseek(2, 0, SEEK_CUR);
            if (end > 0) {
                const off_t from = end > (off_t)sizeof buf2 ? end - (off_t)sizeof buf2 : 0;
                const ssize_t got = pread(2, buf2, (size_t)(end - from), from);
                put(g_fd2, "==== tail of the captured stderr ====\n");
                if (got > 0) { ssize_t r = write(g_fd2, buf2, (size_t)got); (void)r; }
                put(g_fd2, "\n==== end ====\n");
            }
        }
    }
    /* hand on: faulthandler's handler (Python frames), or the default action */
    if ((g_prev.sa_flags & SA_SIGINFO) && g_prev.sa_sigaction) g_prev.sa_sigaction(sig, si, uc);
    else if (!(g_prev.sa_flags & SA_SIGINFO) && g_prev.sa_handler != SIG_DFL && g_prev.sa_handler != SIG_IGN) g_prev.sa_handler(sig);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

int gf3_install_abort_trace(const char *path, int terminal_fd) {
    g_fd2 = terminal_fd;
    void *warm[4];
    (void)backtrace(warm, 4);                  /* loads the unwinder now, not inside the handler */
    g_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, &g_prev);
}
