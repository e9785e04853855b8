/* TEST INFRASTRUCTURE ONLY (oracle/_ref): lets hipcc compile the reference's own CUDA sources, unmodified, from where
 * they lie in /root/reference, so that the reference's kernels can run on the GPU box as the strongest available pin
 * of the oracle.  Nothing here is part of, linked into, or shipped with the product (manigaussian_amd/). */
#pragma once
#include <hip/hip_runtime.h>
#include <iostream>
#include <stdexcept>
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaError_t hipError_t
#define cudaEvent_t hipEvent_t
#define cudaEventCreate hipEventCreate
#define cudaEventRecord hipEventRecord
#define cudaEventSynchronize hipEventSynchronize
#define cudaEventElapsedTime hipEventElapsedTime
/* auxiliary.h:159 */
#define __trap() __builtin_trap()
