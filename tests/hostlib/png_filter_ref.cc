// TEST INFRASTRUCTURE — the oracle of zmx_png_filter_types: LodePNG's own scanline filter search.  `filter` is a static
// function of lodepng.cpp (:5444); this translation unit includes that file where it lies under /root/reference
// (nothing copied) and puts a C symbol in front of it.  Only tests/ loads the resulting library.
#include "lodepng.cpp"

extern "C" __attribute__((visibility("default"))) unsigned ref_png_filter(unsigned char* out, const unsigned char* in, unsigned w, unsigned h,
                                                                           unsigned colortype, unsigned bitdepth, unsigned strategy) {
  LodePNGColorMode mode;
  lodepng_color_mode_init(&mode);
  mode.colortype = (LodePNGColorType)colortype;
  mode.bitdepth = bitdepth;
  LodePNGEncoderSettings settings;
  lodepng_encoder_settings_init(&settings);
  settings.filter_palette_zero = 0;
  settings.filter_strategy = (LodePNGFilterStrategy)strategy;   // LFS_MINSUM = 5, LFS_ENTROPY = 6 (lodepng.h:682-698)
  return filter(out, in, w, h, &mode, &settings);
}
