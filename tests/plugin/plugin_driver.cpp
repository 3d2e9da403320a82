// Drives include/rbt_pcc_plugin.h through the reference's plug-in call pattern (decode a sub-bitstream to a PCCVideo, re-encode it at another QP, keep the
// reconstruction) against the interface double. usage: plugin_driver <in.annexb> <qp> <lossless> <out.annexb> <out_rec.yuv16>
#include "pcc_interface_double.h"
#include "rbt_pcc_plugin.h"
#include <cstring>
int main( int argc, char** argv ) {
  if ( argc < 6 ) return 2;
  using namespace pcc;
  PCCVideoBitstream in, out;
  { FILE* f = fopen( argv[1], "rb" ); if ( !f ) return 2; fseek( f, 0, SEEK_END ); long n = ftell( f ); fseek( f, 0, SEEK_SET ); in.vector().resize( (size_t)n ); if ( fread( in.buffer(), 1, (size_t)n, f ) != (size_t)n ) return 2; fclose( f ); }
  std::shared_ptr<PCCVirtualVideoDecoder<uint16_t>> dec = std::make_shared<PCCRbtVideoDecoder<uint16_t>>();
  std::shared_ptr<PCCVirtualVideoEncoder<uint16_t>> enc = std::make_shared<PCCRbtVideoEncoder<uint16_t>>();
  PCCVideo<uint16_t, 3> video, rec;
  const int lossless = atoi( argv[3] ), bd = lossless ? 8 : 10;
  dec->decode( in, video, (size_t)bd );
  PCCVideoEncoderParameters p; p.qp_ = atoi( argv[2] ); p.inputBitDepth_ = p.internalBitDepth_ = p.outputBitDepth_ = bd; p.transquantBypassEnable_ = lossless != 0;
  enc->encode( video, p, out, rec );
  { FILE* f = fopen( argv[4], "wb" ); fwrite( out.buffer(), 1, out.size(), f ); fclose( f ); }
  { FILE* f = fopen( argv[5], "wb" ); for ( size_t i = 0; i < rec.getFrameCount(); i++ ) for ( size_t c = 0; c < 3; c++ ) { const std::vector<uint16_t>& ch = rec.getFrame( i ).getChannel( c ); fwrite( ch.data(), 2, ch.size(), f ); } fclose( f ); }
  printf( "plugin driver: %zu frames %zux%zu, %zu -> %zu bytes\n", video.getFrameCount(), video.getWidth(), video.getHeight(), in.size(), out.size() );
  return 0;
}
