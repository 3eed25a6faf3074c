// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never linked into or loaded by the product.
//
// The reference's `uniform` dequantisation (reference src/uniform_compression.cpp:64-72: a
// std::transform over the row's bytes with the lambda `min + (max - min) * float(v) / levels`,
// `levels` a uint8_t as flatbuffers hands it out) as a translation unit of its own, compiled the way
// the reference compiles that file -- g++ -std=c++14 -O3 -Wall -Werror for baseline x86-64
// (reference CMakeLists.txt:15: no -march, no -ffast-math, so SSE2 scalar sub / mul / cvtsi2ss / div /
// add, no FMA). The reference's own file cannot be compiled here (it needs the flatc-generated
// headers), so this restated expression, built with the reference's flags, is what pins the fp32
// results bit for bit: tests/golden/uniform_expr.json is generated from it
// (tests/golden/make_golden.py), and oracle/memb_oracle.c's C form and the HIP kernel are checked
// against it.
#include <algorithm>
#include <cstddef>
#include <cstdint>

extern "C" void memb_uniform_expr(
    float minValue, float maxValue, std::uint8_t quantizationLevels, const std::uint8_t* values, std::size_t count,
    float* destination)
{
    std::transform(
        values, values + count, destination,
        [minValue, maxValue, quantizationLevels](std::uint8_t value)
        {
            auto floatValue = static_cast<float>(value);
            return minValue + (maxValue - minValue) * floatValue / quantizationLevels;
        });
}
