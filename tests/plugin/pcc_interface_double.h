// TEST DOUBLE, not reference code: the handful of members of the reference's carrier types that include/rbt_pcc_plugin.h touches, with the
// signatures the reference declares (cited per member), so that the adapter templates can be compiled and run in a container where the reference's
// PccLibCommon cannot be built (it needs a cmake-generated PCCConfig.h). Behaviour is the obvious one (containers of samples); nothing
// of the reference's implementation is reproduced. In the reference tree the adapter is compiled against the real headers instead.
#pragma once
#include <cstdint>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
namespace pcc {
enum class PCCCOLORFORMAT { UNKNOWN = 0, RGB444 = 1, YUV444 = 2, YUV420 = 3 };                       // PCCCommon.h (colour formats of PCCImage)
class PCCVideoBitstream {                                                                            // PCCVideoBitstream.h:46-49
 public:
  std::vector<uint8_t>& vector() { return data_; }
  uint8_t* buffer() { return data_.data(); }
  size_t size() { return data_.size(); }
 private:
  std::vector<uint8_t> data_;
};
template <class T, size_t N> class PCCImage {                                                        // PCCImage.h:63-131
 public:
  size_t getWidth() const { return w_; }
  size_t getHeight() const { return h_; }
  const std::vector<T>& getChannel( size_t i ) const { return ch_[i]; }
  std::vector<T>& getChannel( size_t i ) { return ch_[i]; }
  void resize( size_t w, size_t h, PCCCOLORFORMAT f ) { w_ = w; h_ = h; fmt_ = f; ch_[0].assign( w * h, 0 ); const size_t c = f == PCCCOLORFORMAT::YUV420 ? w * h / 4 : w * h; ch_[1].assign( c, 0 ); ch_[2].assign( c, 0 ); }
  template <typename Pel> void set( const Pel* Y, const Pel* U, const Pel* V, size_t widthY, size_t heightY, size_t strideY, size_t widthC, size_t heightC, size_t strideC,
                                    int16_t shiftbits, PCCCOLORFORMAT format, bool /*rgb2bgr*/ ) {     // PCCImage.h:90-131: copies the planes, right-shifting with rounding
    resize( widthY, heightY, format );
    const Pel* src[3] = {Y, U, V};
    for ( size_t c = 0; c < 3; c++ ) { const size_t w = c ? widthC : widthY, h = c ? heightC : heightY, st = c ? strideC : strideY;
      for ( size_t y = 0; y < h; y++ ) for ( size_t x = 0; x < w; x++ ) { int v = (int)src[c][y * st + x]; if ( shiftbits > 0 ) v = ( v + ( 1 << ( shiftbits - 1 ) ) ) >> shiftbits; ch_[c][y * w + x] = (T)v; } }
  }
 private:
  size_t w_ = 0, h_ = 0; PCCCOLORFORMAT fmt_ = PCCCOLORFORMAT::UNKNOWN; std::vector<T> ch_[N];
};
template <class T, size_t N> class PCCVideo {                                                        // PCCVideo.h:48-83
 public:
  void resize( const size_t frameCount ) { frames_.resize( frameCount ); }
  void clear() { frames_.clear(); }
  PCCImage<T, N>& getFrame( const size_t i ) { return frames_[i]; }
  size_t getWidth() const { return frames_.empty() ? 0 : frames_[0].getWidth(); }
  size_t getHeight() const { return frames_.empty() ? 0 : frames_[0].getHeight(); }
  size_t getFrameCount() const { return frames_.size(); }
 private:
  std::vector<PCCImage<T, N>> frames_;
};
template <class T> class PCCVirtualVideoDecoder {                                                    // PCCVirtualVideoDecoder.h:43-58
 public:
  virtual ~PCCVirtualVideoDecoder() {}
  virtual void decode( PCCVideoBitstream& bitstream, PCCVideo<T, 3>& video, size_t outputBitDepth = 8, const std::string& decoderPath = "", const std::string& parameters = "" ) = 0;
};
struct PCCVideoEncoderParameters {                                                                   // PCCVirtualVideoEncoder.h:42-64 (the fields the adapter reads)
  int32_t qp_ = 30, inputBitDepth_ = 8, internalBitDepth_ = 8, outputBitDepth_ = 8; bool transquantBypassEnable_ = false;
};
template <class T> class PCCVirtualVideoEncoder {                                                    // PCCVirtualVideoEncoder.h:66-86
 public:
  virtual ~PCCVirtualVideoEncoder() {}
  virtual void encode( PCCVideo<T, 3>& videoSrc, PCCVideoEncoderParameters& params, PCCVideoBitstream& bitstream, PCCVideo<T, 3>& videoRec ) = 0;
};
}  // namespace pcc
