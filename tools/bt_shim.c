/* LD_PRELOAD shim for crash hunting on the GPU box: on SIGSEGV / SIGABRT print the faulting address, the instruction
 * pointer, the return addresses found by scanning the stack top for values inside executable mappings, and
 * /proc/self/maps — with nothing but write(2) (a damaged heap kills backtrace()).  Test tooling only; x86-64 Linux only.
 *   gcc -O1 -shared -fPIC -o tools/bt_shim.so tools/bt_shim.c ;  LD_PRELOAD=$PWD/tools/bt_shim.so python3 ... */
#define _GNU_SOURCE
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>
static void hex(const char *tag, uint64_t v) {
  char b[96]; int n = 0;
  while (tag[n]) { b[n] = tag[n]; n++; }
  for (int s = 60; s >= 0; s -= 4) b[n++] = "0123456789abcdef"[(v >> s) & 15];
  b[n++] = '\n';
  (void)!write(2, b, (size_t)n);
}
static void handler(int sig, siginfo_t *si, void *uc_) {
  ucontext_t *uc = (ucontext_t *)uc_;
  hex("[bt_shim] signal ", (uint64_t)sig);
  hex("[bt_shim] fault address ", (uint64_t)(uintptr_t)si->si_addr);
  hex("[bt_shim] rip ", (uint64_t)uc->uc_mcontext.gregs[REG_RIP]);
  /* (the scan stays inside the page-aligned 8 KiB above RSP that a live stack always has mapped, and the handler was
   *  installed with SA_RESETHAND | SA_NODEFER: a fault in here ends the process with the default action, after the
   *  lines above are out) */
  const uint64_t *sp = (const uint64_t *)uc->uc_mcontext.gregs[REG_RSP];
  const uint64_t *lim = (const uint64_t *)((((uintptr_t)sp) | 4095u) + 1 + 4096);
  for (int i = 0; i < 400 && sp + i < lim; i++) { const uint64_t v = sp[i]; if ((v >> 44) == 0x7 || (v >> 44) == 0x5) hex("[bt_shim] stack ", v); }
  int fd = open("/proc/self/maps", O_RDONLY);
  if (fd >= 0) { char buf[4096]; ssize_t k; while ((k = read(fd, buf, sizeof(buf))) > 0) (void)!write(2, buf, (size_t)k); close(fd); }
  _exit(128 + sig);
}
__attribute__((constructor)) static void init(void) {
  struct sigaction sa; memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler; sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_NODEFER;
  sigaction(SIGSEGV, &sa, 0); sigaction(SIGABRT, &sa, 0);
}
