// mggcn_oracle_init.cpp -- TEST INFRASTRUCTURE ONLY (see mggcn_oracle.c header).
//
// dn_matrix::init(gain) -- /root/reference/src/matrix.hpp:539-545:
//   std::default_random_engine gen(99); gain *= sqrt(3.0 / N);
//   uniform_real_distribution uni(-gain, gain); buffer[i] = uni(gen), row-major.
// The engine and the distribution are libstdc++ facilities (minstd_rand0 +
// generate_canonical<float,24>), so the restatement *is* a call into the same
// standard library, compiled with g++ here.  gain arithmetic keeps the
// reference's float/double mix: gain is a float parameter, the sqrt is double,
// the product is narrowed back to float.
#include <cmath>
#include <cstddef>
#include <random>

extern "C" __attribute__((visibility("default")))
void orc_init_uniform(float *buffer, std::size_t n_rows, std::size_t n_cols, float gain) {
    std::default_random_engine gen(99);
    gain *= std::sqrt(3.0 / n_rows);
    std::uniform_real_distribution<float> uni(-gain, gain);
    for (std::size_t i = 0; i < n_rows * n_cols; i++) buffer[i] = uni(gen);
}

// Default gains used by linear::linear -- /root/reference/src/gcn.hpp:107-110:
//   W.init()                      -> gain = sqrt(2 / (1 + 0.01*0.01))  (matrix.hpp:539)
//   b.init(sqrt((r_t)1.0 / 3))    -> gain = sqrt(1/3)
extern "C" __attribute__((visibility("default")))
float orc_default_gain_w(void) { return (float)std::sqrt(2 / (1 + 0.01 * 0.01)); }

extern "C" __attribute__((visibility("default")))
float orc_default_gain_b(void) { return std::sqrt((float)1.0 / 3); }
