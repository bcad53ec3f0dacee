/*
 * libzopflipng_amd.so — zopflipng's optimiser library with the MI355X doing the deflate
 * (ZopfliDeflate of libzopfli_amd.so) and the per-row filter search (zmx_png_filter_types), the filter-strategy
 * trials of one image running side by side on host threads.
 *
 * This header is the ABI of the reference's src/zopflipng/zopflipng_lib.h (:38-83 the C API, :88-144 the C++ API):
 * the same names, enumerators, struct layouts and signatures, so that src/zopflipng/zopflipng_bin.cc — or any program
 * written against zopflipng_lib.h — links against libzopflipng_amd.so instead of zopflipng_lib.cc (INTEGRATION.md
 * section 3).  Results are the reference's PNG files byte for byte (tests/test_gpu_png.py).
 */
#ifndef ZOPFLIPNG_AMD_H_
#define ZOPFLIPNG_AMD_H_

#ifdef __cplusplus
#include <string>
#include <vector>
extern "C" {
#endif

#include <stdlib.h>

/* zopflipng_lib.h:38-49 */
enum ZopfliPNGFilterStrategy {
  kStrategyZero = 0,
  kStrategyOne = 1,
  kStrategyTwo = 2,
  kStrategyThree = 3,
  kStrategyFour = 4,
  kStrategyMinSum,
  kStrategyEntropy,
  kStrategyPredefined,
  kStrategyBruteForce,
  kNumFilterStrategies
};

/* zopflipng_lib.h:51-72 */
typedef struct CZopfliPNGOptions {
  int lossy_transparent;
  int lossy_8bit;
  enum ZopfliPNGFilterStrategy* filter_strategies;
  int num_filter_strategies;
  int auto_filter_strategy;
  char** keepchunks;
  int num_keepchunks;
  int use_zopfli;
  int num_iterations;
  int num_iterations_large;
  int block_split_strategy;
} CZopfliPNGOptions;

/* zopflipng_lib.h:76: the defaults; keepchunks and filter_strategies are neither allocated nor set */
void CZopfliPNGSetDefaults(CZopfliPNGOptions* png_options);

/* zopflipng_lib.h:80-85: 0 on success, an error code otherwise; the caller frees *resultpng */
int CZopfliPNGOptimize(const unsigned char* origpng, const size_t origpng_size, const CZopfliPNGOptions* png_options,
                       int verbose, unsigned char** resultpng, size_t* resultpng_size);

#ifdef __cplusplus
}  /* extern "C" */

/* zopflipng_lib.h:93-137 */
struct ZopfliPNGOptions {
  ZopfliPNGOptions();
  bool verbose;
  bool lossy_transparent;
  bool lossy_8bit;
  std::vector<ZopfliPNGFilterStrategy> filter_strategies;
  bool auto_filter_strategy;
  bool keep_colortype;
  std::vector<std::string> keepchunks;
  bool use_zopfli;
  int num_iterations;
  int num_iterations_large;
  int block_split_strategy;
};

/* zopflipng_lib.h:141-144: 0 if ok */
int ZopfliPNGOptimize(const std::vector<unsigned char>& origpng, const ZopfliPNGOptions& png_options, bool verbose,
                      std::vector<unsigned char>* resultpng);
#endif

#endif  /* ZOPFLIPNG_AMD_H_ */
