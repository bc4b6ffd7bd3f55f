// Debug aid (LD_PRELOAD): print a backtrace at every C++ throw.  gcc -shared -fPIC -o /tmp/throw_trace.so tools/throw_trace.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <unistd.h>
#include <string.h>
typedef void (*throw_fn)(void*, void*, void (*)(void*));
void __cxa_throw(void* obj, void* tinfo, void (*dest)(void*))
{
  static throw_fn real;
  if (!real) real = (throw_fn)dlsym(RTLD_NEXT, "__cxa_throw");
  if (!real) { void* h = dlopen("libstdc++.so.6", RTLD_LAZY); if (h) real = (throw_fn)dlsym(h, "__cxa_throw"); }
  void* bt[48];
  const int n = backtrace(bt, 48);
  const char* msg = "---- throw ----\n";
  (void)!write(2, msg, strlen(msg));
  backtrace_symbols_fd(bt, n, 2);
  real(obj, tinfo, dest);
  __builtin_unreachable();
}
