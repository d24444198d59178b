// Error channel and device queries of libswem_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void swem_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int swem_version(void) { return SWEM_ABI_VERSION; }
extern "C" const char *swem_last_error(void) { return g_err; }
extern "C" int swem_device_cus(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return cus;
}
