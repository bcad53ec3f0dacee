// libzopflipng_amd.so: zopflipng's optimiser on top of libzopfli_amd.so.
//
// What the reference does (src/zopflipng/zopflipng_lib.cc): decode the PNG with LodePNG, optionally clean up
// invisible colours (:96-156), encode the image once per filter strategy with LodePNG's own fast deflate to see which
// strategy gives the smallest file (:270-305, eight encodes one after the other), encode it again with that strategy
// and Zopfli as the deflate (:160-268, :430-470), copy the chunks the caller wants to keep (:325-352).
//
// What this file does with the same inputs, for the same bytes out:
//   * the trials of an image run SIDE BY SIDE, a host thread each: they share nothing but the decoded pixels;
//   * the per-row filter search of the MINSUM and ENTROPY strategies (lodepng.cpp:5444-5570) runs on the device
//     (zmx_png_filter_types): the raw scanlines LodePNG filters are taken from one uncompressed encode of the image
//     (LodePNG chooses the colour model itself, lodepng.cpp:5851-5925: this way the scanlines are its, whatever it
//     chooses), the device returns the filter type of every row for both strategies, and the trial (and the final
//     encode) hands them to LodePNG as LFS_PREDEFINED — the scanlines it writes are the ones its own search would
//     have written;
//   * the final deflate is ZopfliDeflate of libzopfli_amd.so (the LZ77 optimal parse on the MI355X).
// LodePNG itself (decoder, colour conversion, chunk writer, the trials' deflate) is the third-party library the
// reference vendors; it is compiled from wherever LODEPNG_DIR points (zopfli_amd/_build.py), not part of this source.
#include "zopflipng_amd.h"

#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <set>
#include <thread>
#include <unordered_set>

#include "lodepng.h"
#include "lodepng_util.h"
#include "zopfli_amd.h"

ZopfliPNGOptions::ZopfliPNGOptions()      // zopflipng_lib.cc:33-43
    : verbose(false), lossy_transparent(false), lossy_8bit(false), auto_filter_strategy(true), keep_colortype(false),
      use_zopfli(true), num_iterations(15), num_iterations_large(5), block_split_strategy(1) {}

namespace {

// LodePNG's custom_deflate hook (lodepng.h LodePNGCompressSettings): zopflipng_lib.cc:47-66
unsigned DeflateOnDevice(unsigned char** out, size_t* outsize, const unsigned char* in, size_t insize,
                         const LodePNGCompressSettings* settings) {
  const ZopfliPNGOptions* o = static_cast<const ZopfliPNGOptions*>(settings->custom_context);
  ZopfliOptions z;
  ZopfliInitOptions(&z);
  z.verbose = o->verbose;
  z.numiterations = insize < 200000 ? o->num_iterations : o->num_iterations_large;
  unsigned char bp = 0;
  ZopfliDeflate(&z, 2, 1, in, insize, &bp, out, outsize);
  return 0;
}

inline unsigned Rgba(const unsigned char* p) { return p[0] + 256u * p[1] + 65536u * p[2] + 16777216u * p[3]; }

// the distinct colours of an RGBA8 image, counting stops beyond 256 (zopflipng_lib.cc:72-83)
void DistinctColors(std::unordered_set<unsigned>* seen, const unsigned char* px, size_t n, bool transparent_as_one) {
  seen->clear();
  for (size_t i = 0; i < n; ++i) {
    const unsigned char* p = px + 4 * i;
    seen->insert(transparent_as_one && p[3] == 0 ? 0u : Rgba(p));
    if (seen->size() > 256) break;
  }
}

// zopflipng_lib.cc:87-156: fully transparent pixels get a colour that filters well
void CleanInvisibleColors(lodepng::State* inputstate, unsigned char* px, unsigned w, unsigned h) {
  const size_t n = static_cast<size_t>(w) * h;
  bool key = true;        // no translucent pixel: a colour key can stand for the transparency
  for (size_t i = 0; i < n && key; ++i) key = px[4 * i + 3] == 0 || px[4 * i + 3] == 255;
  std::unordered_set<unsigned> seen;
  DistinctColors(&seen, px, n, true);
  const bool palette = seen.size() <= 256;
  unsigned char fill[3] = {0, 0, 0};
  if (key || palette) {   // the first transparent pixel's colour: a valid key / a colour the palette has
    for (size_t i = 0; i < n; ++i) {
      if (px[4 * i + 3] == 0) { memcpy(fill, px + 4 * i, 3); break; }
    }
  }
  for (size_t i = 0; i < n; ++i) {
    unsigned char* p = px + 4 * i;
    if (p[3] == 0) memcpy(p, fill, 3);
    else if (!key && !palette) memcpy(fill, p, 3);     // the last visible colour: zeros for the PNG filters
  }
  LodePNGColorMode& c = inputstate->info_png.color;
  if (palette && c.palettesize > 0) {
    DistinctColors(&seen, px, n, false);
    if (seen.size() < c.palettesize) {                // colours went away: the input palette without them, in its order
      size_t kept = 0;
      for (size_t i = 0; i < c.palettesize; ++i) {
        if (seen.count(Rgba(c.palette + 4 * i))) {
          if (kept != i) memmove(c.palette + 4 * kept, c.palette + 4 * i, 4);
          ++kept;
        }
      }
      c.palettesize = kept;
    }
  }
}

// One image, ready to be encoded under any filter strategy.
class Image {
 public:
  std::vector<unsigned char> pixels;
  unsigned w = 0, h = 0;
  lodepng::State input;           // the input file's header and colour mode
  bool bit16 = false, keep_colortype = false;
  const std::vector<unsigned char>* origfile = nullptr;

  // zopflipng_lib.cc:160-268 with `strategy`; the MINSUM / ENTROPY row search from the device when it is at hand
  unsigned Encode(ZopfliPNGFilterStrategy strategy, bool zopfli, int windowsize, const ZopfliPNGOptions* options,
                  std::vector<unsigned char>* out) const {
    lodepng::State state;
    Configure(&state, windowsize);
    if (zopfli && options->use_zopfli) {
      state.encoder.zlibsettings.custom_deflate = DeflateOnDevice;
      state.encoder.zlibsettings.custom_context = options;
    }
    std::vector<unsigned char> filters;
    LodePNGFilterStrategy lfs = LFS_ZERO;
    switch (strategy) {
      case kStrategyZero: lfs = LFS_ZERO; break;
      case kStrategyOne: lfs = LFS_ONE; break;
      case kStrategyTwo: lfs = LFS_TWO; break;
      case kStrategyThree: lfs = LFS_THREE; break;
      case kStrategyFour: lfs = LFS_FOUR; break;
      case kStrategyMinSum: lfs = LFS_MINSUM; break;
      case kStrategyEntropy: lfs = LFS_ENTROPY; break;
      case kStrategyBruteForce: lfs = LFS_BRUTE_FORCE; break;
      case kStrategyPredefined:
        lodepng::getFilterTypes(filters, *origfile);
        if (filters.size() != h) return 1;
        lfs = LFS_PREDEFINED;
        break;
      default: lfs = state.encoder.filter_strategy; break;
    }
    state.encoder.filter_strategy = lfs;
    if (lfs == LFS_PREDEFINED) state.encoder.predefined_filters = filters.data();
    const std::vector<unsigned char>* searched = lfs == LFS_MINSUM ? &minsum_ : lfs == LFS_ENTROPY ? &entropy_ : nullptr;
    if (searched && searched->size() == h) {          // the device's search: the same rows, as predefined types
      state.encoder.filter_strategy = LFS_PREDEFINED;
      state.encoder.predefined_filters = searched->data();
    }
    unsigned error = lodepng::encode(*out, pixels, w, h, state);
    // zopflipng_lib.cc:232-258: a very small file may be smaller without its palette
    if (!error && out->size() < 4096 && !keep_colortype && lodepng::getPNGHeaderInfo(*out).color.colortype == LCT_PALETTE) {
      LodePNGColorStats stats;
      lodepng_color_stats_init(&stats);
      lodepng_compute_color_stats(&stats, pixels.data(), w, h, &state.info_raw);
      if (w * h <= 16 && stats.key) stats.alpha = 1;    // too small for the tRNS chunk's overhead
      state.encoder.auto_convert = 0;
      state.info_png.color.colortype = stats.alpha ? LCT_RGBA : LCT_RGB;
      state.info_png.color.bitdepth = 8;
      state.info_png.color.key_defined = stats.key && !stats.alpha;
      if (state.info_png.color.key_defined) {
        state.info_png.color.key_defined = 1;
        state.info_png.color.key_r = stats.key_r & 255u;
        state.info_png.color.key_g = stats.key_g & 255u;
        state.info_png.color.key_b = stats.key_b & 255u;
      }
      // (other scanlines than the device searched: LodePNG's own search here)
      state.encoder.filter_strategy = lfs;
      state.encoder.predefined_filters = lfs == LFS_PREDEFINED ? filters.data() : nullptr;
      std::vector<unsigned char> out2;
      error = lodepng::encode(out2, pixels, w, h, state);
      if (out2.size() < out->size()) out->swap(out2);
    }
    if (error) printf("Encoding error %u: %s\n", error, lodepng_error_text(error));
    return error;
  }

  // The filter type of every scanline under MINSUM and under ENTROPY, searched on the device.  The scanlines are
  // LodePNG's: one encode with filter type 0 and stored deflate blocks is the raw (colour-converted, bit-padded) rows
  // behind a zero byte each.  An image LodePNG would lay out differently than assumed here (interlaced; a header that
  // does not match) leaves the vectors empty and the rows to LodePNG's own search — the same rows by definition.  A
  // DEVICE failure is not papered over that way: it is an error of the optimisation (returns false, message on
  // stderr), like a failing ZopfliDeflate.
  bool SearchFiltersOnDevice() {
    minsum_.clear();
    entropy_.clear();
    if (getenv("ZOPFLIPNG_AMD_HOST_FILTERS")) return true;     // (A/B and test hook)
    lodepng::State state;
    Configure(&state, 32768);
    state.encoder.filter_strategy = LFS_ZERO;
    state.encoder.zlibsettings.btype = 0;
    std::vector<unsigned char> png;
    if (lodepng::encode(png, pixels, w, h, state) != 0) return true;
    lodepng::State hdr;
    unsigned pw = 0, ph = 0;
    if (lodepng_inspect(&pw, &ph, &hdr, png.data(), png.size()) != 0 || pw != w || ph != h) return true;
    if (hdr.info_png.interlace_method != 0) return true;
    const unsigned bpp = lodepng_get_bpp(&hdr.info_png.color);
    if (bpp == 0) return true;
    const size_t linebytes = (static_cast<size_t>(w) * bpp + 7) / 8, bytewidth = (bpp + 7) / 8;
    std::vector<unsigned char> idat;
    for (const unsigned char* c = png.data() + 8; c + 12 <= png.data() + png.size(); c = lodepng_chunk_next_const(c, png.data() + png.size())) {
      if (lodepng_chunk_type_equals(c, "IDAT")) idat.insert(idat.end(), lodepng_chunk_data_const(c), lodepng_chunk_data_const(c) + lodepng_chunk_length(c));
      if (lodepng_chunk_type_equals(c, "IEND")) break;
    }
    std::vector<unsigned char> rows;
    if (lodepng::decompress(rows, idat) != 0 || rows.size() != static_cast<size_t>(h) * (linebytes + 1)) return true;
    std::vector<unsigned char> raw(static_cast<size_t>(h) * linebytes);
    for (size_t y = 0; y < h; ++y) memcpy(raw.data() + y * linebytes, rows.data() + y * (linebytes + 1) + 1, linebytes);
    std::vector<unsigned char> a(h), b(h);
    if (zmx_png_filter_types_pooled(raw.data(), linebytes, h, bytewidth, a.data(), b.data()) != 0) {
      fprintf(stderr, "zopflipng_amd: the row-filter search on the device failed: %s\n", zmx_last_error());
      return false;
    }
    minsum_.swap(a);
    entropy_.swap(b);
    return true;
  }

 private:
  // the settings every encode of this image shares (zopflipng_lib.cc:169-192, :225-226)
  void Configure(lodepng::State* state, int windowsize) const {
    state->encoder.zlibsettings.windowsize = windowsize;
    if (keep_colortype) {
      state->encoder.auto_convert = 0;
      lodepng_color_mode_copy(&state->info_png.color, &input.info_png.color);
    }
    if (input.info_png.color.colortype == LCT_PALETTE) {      // the palette in its original order
      lodepng_color_mode_copy(&state->info_raw, &input.info_png.color);
      state->info_raw.colortype = LCT_RGBA;
      state->info_raw.bitdepth = 8;
    }
    if (bit16) state->info_raw.bitdepth = 16;
    state->encoder.filter_palette_zero = 0;
    state->encoder.add_id = false;
    state->encoder.text_compression = 1;
  }
  std::vector<unsigned char> minsum_, entropy_;
};

// chunk names of `keep` that the file has, per place (before PLTE, before IDAT, after IDAT): zopflipng_lib.cc:309-352
void NamesPresent(const std::vector<unsigned char>& png, const std::vector<std::string>& keep, std::set<std::string>* out) {
  std::vector<std::string> names[3];
  std::vector<std::vector<unsigned char> > chunks[3];
  lodepng::getChunks(names, chunks, png);
  for (int place = 0; place < 3; ++place)
    for (const std::string& n : names[place])
      if (std::find(keep.begin(), keep.end(), n) != keep.end()) out->insert(n);
}
void CopyKeptChunks(const std::vector<unsigned char>& from, const std::vector<std::string>& keep, std::vector<unsigned char>* png) {
  std::vector<std::string> names[3];
  std::vector<std::vector<unsigned char> > chunks[3], kept[3];
  lodepng::getChunks(names, chunks, from);
  for (int place = 0; place < 3; ++place)
    for (size_t j = 0; j < names[place].size(); ++j)
      for (const std::string& k : keep)
        if (k == names[place][j]) kept[place].push_back(chunks[place][j]);
  lodepng::insertChunks(*png, kept);
}

const char* const kStrategyName[kNumFilterStrategies] = {"zero", "one", "two", "three", "four", "minimum sum", "entropy", "predefined", "brute force"};

}  // namespace

int ZopfliPNGOptimize(const std::vector<unsigned char>& origpng, const ZopfliPNGOptions& png_options, bool verbose,
                      std::vector<unsigned char>* resultpng) {
  bool enable[kNumFilterStrategies] = {false, false, false, false, false, false, false, false, false};
  for (ZopfliPNGFilterStrategy s : png_options.filter_strategies) enable[s] = true;

  Image img;
  img.origfile = &origpng;
  unsigned error = lodepng::decode(img.pixels, img.w, img.h, img.input, origpng);
  img.keep_colortype = png_options.keep_colortype;
  if (!png_options.keepchunks.empty()) {
    // bKGD and sBIT are written in terms of the colour type: keeping them keeps the type (zopflipng_lib.cc:387-404)
    std::set<std::string> present;
    NamesPresent(origpng, png_options.keepchunks, &present);
    if (present.count("bKGD") || present.count("sBIT")) {
      if (!img.keep_colortype && verbose) printf("Forced to keep original color type due to keeping bKGD or sBIT chunk.\n");
      img.keep_colortype = true;
    }
  }
  if (error) {
    if (verbose) {
      if (error == 1) printf("Decoding error\n");
      else printf("Decoding error %u: %s\n", error, lodepng_error_text(error));
    }
    return static_cast<int>(error);
  }
  if (img.input.info_png.color.bitdepth == 16 && (img.keep_colortype || !png_options.lossy_8bit)) {
    img.pixels.clear();
    error = lodepng::decode(img.pixels, img.w, img.h, origpng, LCT_RGBA, 16);
    img.bit16 = true;
  }
  if (!error && png_options.lossy_transparent && !img.bit16) CleanInvisibleColors(&img.input, img.pixels.data(), img.w, img.h);

  if (!error) {
    // which strategies will be encoded at all decides whether the device's row search is wanted
    const bool trials = png_options.auto_filter_strategy;
    if ((trials || enable[kStrategyMinSum] || enable[kStrategyEntropy]) && !img.SearchFiltersOnDevice()) error = 1;
    if (trials && !error) {
      // zopflipng_lib.cc:270-305: every strategy but brute force with LodePNG's fast deflate (window 8192: the winner
      // depends on the window), the smallest file's strategy wins (the first of equals) — the encodes side by side
      const int n = kNumFilterStrategies - 1;
      std::vector<std::vector<unsigned char> > out(n);
      std::vector<unsigned> err(n, 0);
      std::vector<std::thread> workers;
      for (int i = 0; i < n; ++i) {
        workers.emplace_back([&, i] { err[i] = img.Encode(static_cast<ZopfliPNGFilterStrategy>(i), false, 8192, nullptr, &out[i]); });
      }
      for (auto& t : workers) t.join();
      size_t bestsize = 0;
      int best = 0;
      for (int i = 0; i < n && !error; ++i) {
        if (err[i]) { error = err[i]; break; }
        if (bestsize == 0 || out[i].size() < bestsize) { bestsize = out[i].size(); best = i; }
      }
      if (!error) for (int i = 0; i < n; ++i) enable[i] = i == best;
    }
  }

  if (!error) {
    size_t bestsize = 0;
    for (int i = 0; i < kNumFilterStrategies; ++i) {
      if (!enable[i]) continue;
      std::vector<unsigned char> temp;
      error = img.Encode(static_cast<ZopfliPNGFilterStrategy>(i), true, 32768, &png_options, &temp);
      if (!error) {
        if (verbose) printf("Filter strategy %s: %d bytes\n", kStrategyName[i], static_cast<int>(temp.size()));
        if (bestsize == 0 || temp.size() < bestsize) {
          bestsize = temp.size();
          resultpng->swap(temp);
        }
      }
    }
    if (!png_options.keepchunks.empty()) CopyKeptChunks(origpng, png_options.keepchunks, resultpng);
  }
  return static_cast<int>(error);
}

extern "C" void CZopfliPNGSetDefaults(CZopfliPNGOptions* png_options) {
  memset(png_options, 0, sizeof(*png_options));
  const ZopfliPNGOptions d;
  png_options->lossy_transparent = d.lossy_transparent;
  png_options->lossy_8bit = d.lossy_8bit;
  png_options->auto_filter_strategy = d.auto_filter_strategy;
  png_options->use_zopfli = d.use_zopfli;
  png_options->num_iterations = d.num_iterations;
  png_options->num_iterations_large = d.num_iterations_large;
  png_options->block_split_strategy = d.block_split_strategy;
}

extern "C" int CZopfliPNGOptimize(const unsigned char* origpng, const size_t origpng_size, const CZopfliPNGOptions* png_options,
                                  int verbose, unsigned char** resultpng, size_t* resultpng_size) {
  ZopfliPNGOptions o;
  o.lossy_transparent = png_options->lossy_transparent != 0;
  o.lossy_8bit = png_options->lossy_8bit != 0;
  o.auto_filter_strategy = png_options->auto_filter_strategy != 0;
  o.use_zopfli = png_options->use_zopfli != 0;
  o.num_iterations = png_options->num_iterations;
  o.num_iterations_large = png_options->num_iterations_large;
  o.block_split_strategy = png_options->block_split_strategy;
  o.filter_strategies.assign(png_options->filter_strategies, png_options->filter_strategies + png_options->num_filter_strategies);
  for (int i = 0; i < png_options->num_keepchunks; ++i) o.keepchunks.push_back(png_options->keepchunks[i]);
  const std::vector<unsigned char> in(origpng, origpng + origpng_size);
  std::vector<unsigned char> out;
  const int rc = ZopfliPNGOptimize(in, o, verbose != 0, &out);
  if (rc) return rc;
  *resultpng_size = out.size();
  *resultpng = static_cast<unsigned char*>(malloc(out.size()));
  if (!*resultpng) return ENOMEM;
  memcpy(*resultpng, out.data(), out.size());
  return 0;
}
