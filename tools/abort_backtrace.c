/* LD_PRELOAD shim for hunting an abort() somewhere below Python: writes the aborting thread's backtrace to
 * $ABORT_BT_FILE (default abort_bt.txt) before the default action. Build: gcc -shared -fPIC -o abort_bt.so abort_backtrace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_abort(int sig) {
    void* bt[96];
    const int n = backtrace(bt, 96);
    const char* path = getenv("ABORT_BT_FILE");
    const int fd = open(path ? path : "abort_bt.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        const char* m = "---- SIGABRT backtrace ----\n";
        if (write(fd, m, strlen(m)) < 0) {}
        backtrace_symbols_fd(bt, n, fd);
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_abort); }
