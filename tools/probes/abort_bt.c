/* LD_PRELOAD helper: print a native backtrace to stderr when the process receives SIGABRT (a HIP / ROCr runtime abort leaves no
 * message on this image and core dumps are disabled).   gcc -shared -fPIC -O1 tools/probes/abort_bt.c -o tools/probes/abort_bt.so
 *   LD_PRELOAD=tools/probes/abort_bt.so python -m pytest -p no:faulthandler ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
    void* frames[96];
    const char msg[] = "\n==== SIGABRT: native backtrace ====\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, 0);
}
