// common.h -- error plumbing shared by the host-side translation units of libpsvr_engine.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include "../../include/psvr_engine.h"

namespace psvr {

std::string &last_error_ref();

inline int set_error(int code, const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	last_error_ref() = buf;
	return code;
}

#define PSVR_HIP(call)                                                                        \
	do {                                                                                      \
		hipError_t e_ = (call);                                                               \
		if (e_ != hipSuccess)                                                                 \
			return psvr::set_error(PSVR_ERR_DEVICE, "%s failed: %s (%s:%d)", #call,           \
			                       hipGetErrorString(e_), __FILE__, __LINE__);                \
	} while (0)

// RAII device buffer (plain hipMalloc; sized for 288 GB HBM, no pooling needed at this level)
struct DevBuf {
	void *p = nullptr;
	size_t bytes = 0;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { release(); }
	void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
	hipError_t alloc(size_t n)
	{
		release();
		if (n == 0) n = 16;
		hipError_t e = hipMalloc(&p, n);
		if (e == hipSuccess) bytes = n;
		return e;
	}
	hipError_t ensure(size_t n) { return n <= bytes ? hipSuccess : alloc(n + n / 4); }
	template <class T> T *as() const { return (T *)p; }
};

} // namespace psvr
