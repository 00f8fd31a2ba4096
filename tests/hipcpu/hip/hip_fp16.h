// TEST INFRASTRUCTURE: the two conversions the kernels use, on the host's native half type.
#pragma once
typedef _Float16 __half;
inline float __half2float(__half h) { return (float)h; }
inline __half __float2half(float f) { return (__half)f; }
