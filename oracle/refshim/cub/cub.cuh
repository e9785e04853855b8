// oracle/refshim -- TEST INFRASTRUCTURE ONLY: lets the reference's rasterizer_impl.cu (which includes <cub/cub.cuh>)
// compile with hipcc; hipCUB keeps CUB's DeviceRadixSort / DeviceScan contracts.
#pragma once
// the _f32 variant passes the reference's config.h macros on the command line; hipCUB uses NUM_CHANNELS as an identifier
#pragma push_macro("NUM_CHANNELS")
#undef NUM_CHANNELS
#include <hipcub/hipcub.hpp>
#pragma pop_macro("NUM_CHANNELS")
namespace cub = hipcub;
