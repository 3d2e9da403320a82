// ORACLE — test infrastructure only. Harness around the REFERENCE's own HEVC high-level-syntax parser
// (/root/reference/dependencies/PccLibHevcParser, compiled in place by oracle/ref_build.sh into oracle/_ref/).
// It exists to pin the restatement:
//   tables            -> dumps the normative ROM the reference carries (PccHevcTComRom.cpp:457-465,471-616,635-642,700-709)
//   slices <annexb file> ... -> every slice segment header through the reference's parseSliceHeader (PccHevcTDecCAVLC.cpp:1138)
//   hls <annexb file> -> parses every VPS/SPS/PPS of an Annex-B stream with the reference's TDecCavlc
//                        (PccHevcTDecCAVLC.cpp:189 parsePPS, :651 parseSPS, :1043 parseVPS) and prints the fields
// No reference source is copied: the reference files are compiled from where they lie.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "PccHevcParser.h"
#include "PccHevcTComRom.h"
#include "PccHevcContextTables.h"
using namespace pcc_hevc;
// The reference defines its ROM tables at global scope as `const` (PccHevcTComRom.cpp:99 `using namespace`), i.e. with
// internal linkage, so they cannot be linked against: compile that translation unit as part of this one instead
// (ref_build.sh passes -I<reference>/dependencies/PccLibHevcParser/source and leaves its object out of the link).
#include "PccHevcTComRom.cpp"

// The reference declares calculateParameterSetChangedFlagPccHevc inside namespace pcc_hevc (PccHevcTComSlice.h:1739) but defines it at global scope
// (PccHevcTComSlice.cpp:2492, after a using-directive), so ParameterSetMap::storePS - instantiated here, never in the reference's own code - does not link.
// Forward the namespaced name to the reference's own definition (no behaviour of ours involved).
Void calculateParameterSetChangedFlagPccHevc(Bool& bChanged, const std::vector<UChar>* pOldData, const std::vector<UChar>* pNewData);
namespace pcc_hevc { Void calculateParameterSetChangedFlagPccHevc(Bool& bChanged, const std::vector<UChar>* pOldData, const std::vector<UChar>* pNewData) { ::calculateParameterSetChangedFlagPccHevc(bChanged, pOldData, pNewData); } }

static std::vector<uint8_t> readFile(const char* p) {
  std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) exit(2); fclose(f); return v;
}
template <class T> static void dumpArr(const char* name, const T* p, int n, bool last = false) {
  printf("\"%s\": [", name); for (int i = 0; i < n; i++) printf("%s%d", i ? "," : "", (int)p[i]); printf("]%s\n", last ? "" : ",");
}
#define DUMP_INIT(name) dumpArr(#name, &name[0][0], (int)(sizeof(name) / sizeof(name[0][0])))

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "tables")) {
    printf("{\n");
    dumpArr("T4", &::gPccHevc_aiT4[0][0][0], 16); dumpArr("T8", &::gPccHevc_aiT8[0][0][0], 64);
    dumpArr("T16", &::gPccHevc_aiT16[0][0][0], 256); dumpArr("T32", &::gPccHevc_aiT32[0][0][0], 1024);
    dumpArr("DST4", &::gPccHevc_as_DST_MAT_4[0][0][0], 16);
    dumpArr("quantScales", ::gPccHevc_quantScales, 6); dumpArr("invQuantScales", ::gPccHevc_invQuantScales, 6);
    dumpArr("chromaScale420", ::gPccHevc_aucChromaScale[1], 58);
    dumpArr("ctxIndMap4x4", ::ctxIndMap4x4, 16); dumpArr("groupIdx", ::gPccHevc_uiGroupIdx, 32); dumpArr("minInGroup", ::gPccHevc_uiMinInGroup, 10);
    // CABAC initValues, rows B,P,I (PccHevcContextTables.h:186-573)
    DUMP_INIT(INIT_CU_TRANSQUANT_BYPASS_FLAG); DUMP_INIT(INIT_SPLIT_FLAG); DUMP_INIT(INIT_SKIP_FLAG); DUMP_INIT(INIT_MERGE_FLAG_EXT);
    DUMP_INIT(INIT_MERGE_IDX_EXT); DUMP_INIT(INIT_PART_SIZE); DUMP_INIT(INIT_PRED_MODE); DUMP_INIT(INIT_INTRA_PRED_MODE);
    DUMP_INIT(INIT_CHROMA_PRED_MODE); DUMP_INIT(INIT_INTER_DIR); DUMP_INIT(INIT_MVD); DUMP_INIT(INIT_REF_PIC); DUMP_INIT(INIT_DQP);
    DUMP_INIT(INIT_QT_CBF); DUMP_INIT(INIT_QT_ROOT_CBF); DUMP_INIT(INIT_LAST); DUMP_INIT(INIT_SIG_CG_FLAG); DUMP_INIT(INIT_SIG_FLAG);
    DUMP_INIT(INIT_ONE_FLAG); DUMP_INIT(INIT_ABS_FLAG); DUMP_INIT(INIT_MVP_IDX); DUMP_INIT(INIT_SAO_MERGE_FLAG); DUMP_INIT(INIT_SAO_TYPE_IDX);
    DUMP_INIT(INIT_TRANS_SUBDIV_FLAG); DUMP_INIT(INIT_TRANSFORMSKIP_FLAG);
    // coefficient scan orders (PccHevcTComRom.cpp:137-390): [scan type diag/hor/ver][log2 size 2..5] -> raster positions
    ::initROMPccHevc();
    const int types[3] = {SCAN_DIAG, SCAN_HOR, SCAN_VER}; const char* tn[3] = {"diag", "hor", "ver"};
    for (int t = 0; t < 3; t++) for (int l = 1; l <= 5; l++) {
      char nm[64]; snprintf(nm, sizeof nm, "scan_%s_%d", tn[t], l);
      dumpArr(nm, gPccHevc_scanOrder[SCAN_GROUPED_4x4][types[t]][l][l], 1 << (2 * l));
    }
    printf("\"end\": [0]\n}\n");
    return 0;
  }
  if (argc >= 3 && !strcmp(argv[1], "hls")) {
    std::vector<uint8_t> buf = readFile(argv[2]);
    const int size = (int)buf.size(); const uint8_t* data = buf.data();
    TDecCavlc* dec = new TDecCavlc();
    int index = 0, sc = data[2] == 0 ? 4 : 3, n = 0;
    printf("[\n");
    for (int i = sc; i <= size; i++) {
      if (i == size || (i + 3 < size && data[i] == 0 && data[i + 1] == 0 && ((data[i + 2] == 0 && data[i + 3] == 1) || data[i + 2] == 1))) {
        int type = (data[index + sc] & 126) >> 1;
        // strip emulation prevention before handing the payload to the reference reader
        std::vector<uint8_t> rb; int z = 0;
        for (int k = index + sc + 2; k < i; k++) { if (z >= 2 && data[k] == 3) { z = 0; continue; } z = data[k] == 0 ? z + 1 : 0; rb.push_back(data[k]); }
        dec->setBuffer((UChar*)rb.data(), (int)rb.size());
        if (type == NAL_UNIT_VPS) { TComVPS vps; dec->parseVPS(&vps); printf("%s{\"nal\":\"VPS\",\"max_dec_pic_buffering\":%d,\"num_reorder\":%d}\n", n++ ? "," : "", vps.getMaxDecPicBuffering(0), vps.getNumReorderPics(0)); }
        if (type == NAL_UNIT_SPS) {
          TComSPS* s = dec->getSPS(); dec->parseSPS(s);
          // the reference's parseSPS stops storing fields after bit_depth_chroma (PccHevcTDecCAVLC.cpp:734: rest commented out)
          printf("%s{\"nal\":\"SPS\",\"width\":%d,\"height\":%d,\"bit_depth\":%d,\"bit_depth_c\":%d,\"chroma_format\":%d}\n", n++ ? "," : "",
                 (int)s->getPicWidthInLumaSamples(), (int)s->getPicHeightInLumaSamples(), s->getBitDepth(CHANNEL_TYPE_LUMA), s->getBitDepth(CHANNEL_TYPE_CHROMA), (int)s->getChromaFormatIdc());
        }
        if (type == NAL_UNIT_PPS) {
          TComPPS* p = dec->getPPS(); dec->parsePPS(p);
          printf("%s{\"nal\":\"PPS\",\"init_qp\":%d,\"sign_hiding\":%d,\"cabac_init_present\":%d,\"num_ref_idx_l0\":%d,\"cip\":%d,\"transform_skip\":%d,"
                 "\"cu_qp_delta\":%d,\"tq_bypass\":%d,\"lf_across_slices\":%d,\"deblock_ctrl\":%d,\"deblock_disabled\":%d,\"log2_par_mrg\":%d,\"dependent_slice_segments\":%d,\"entropy_coding_sync\":%d}\n", n++ ? "," : "",
                 26 + p->getPicInitQPMinus26(), (int)p->getSignDataHidingEnabledFlag(), (int)p->getCabacInitPresentFlag(), (int)p->getNumRefIdxL0DefaultActive(),
                 (int)p->getConstrainedIntraPred(), (int)p->getUseTransformSkip(), (int)p->getUseDQP(), (int)p->getTransquantBypassEnabledFlag(),
                 (int)p->getLoopFilterAcrossSlicesEnabledFlag(), (int)p->getDeblockingFilterControlPresentFlag(), (int)p->getPPSDeblockingFilterDisabledFlag(),
                 2 + (int)p->getLog2ParallelMergeLevelMinus2(), (int)p->getDependentSliceSegmentsEnabledFlag(), (int)p->getEntropyCodingSyncEnabledFlag());
        }
        if (i < size) { sc = data[i + 2] == 0 ? 4 : 3; index = i; i += sc; }
      }
    }
    printf("]\n");
    delete dec;
    return 0;
  }
  if (argc >= 8 && !strcmp(argv[1], "slices")) {
    // slices <annexb> <log2_max_poc_lsb> <log2_ctb> <sao> <tmvp> <num_st_rps>: every slice segment header of the stream through the reference's
    // TDecCavlc::parseSliceHeader (PccHevcTDecCAVLC.cpp:1138). The reference's parseSPS stops storing fields after the bit depths (:732, the rest is
    // commented out), so the SPS fields the slice header syntax depends on are set from the command line (the caller's own parse of the same SPS);
    // short-term RPS i of the SPS = the i + 1 preceding pictures, all used (what oracle/hevc_enc.c writes). The PPS is the reference's own parse.
    std::vector<uint8_t> buf = readFile(argv[2]);
    const int bitsPoc = atoi(argv[3]), log2Ctb = atoi(argv[4]), sao = atoi(argv[5]), tmvp = atoi(argv[6]), nRps = atoi(argv[7]);
    const int ctcSets = argc >= 9 ? atoi(argv[8]) : 0;   // 1: the sets of the CTC structure (oracle_enc_params.ctc_gop): {-1}, {-2}, {-1,-2}
    const int size = (int)buf.size(); const uint8_t* data = buf.data();
    TDecCavlc* dec = new TDecCavlc();
    ParameterSetManager psm;
    int index = 0, sc = data[2] == 0 ? 4 : 3, n = 0, prevPoc = 0;
    printf("[\n");
    for (int i = sc; i <= size; i++) {
      if (i == size || (i + 3 < size && data[i] == 0 && data[i + 1] == 0 && ((data[i + 2] == 0 && data[i + 3] == 1) || data[i + 2] == 1))) {
        int type = (data[index + sc] & 126) >> 1;
        std::vector<uint8_t> rb; int z = 0;
        for (int k = index + sc + 2; k < i; k++) { if (z >= 2 && data[k] == 3) { z = 0; continue; } z = data[k] == 0 ? z + 1 : 0; rb.push_back(data[k]); }
        dec->setBuffer((UChar*)rb.data(), (int)rb.size());
        std::vector<UChar> nalu(rb.begin(), rb.end());
        if (type == NAL_UNIT_VPS) { TComVPS* vps = new TComVPS(); dec->parseVPS(vps); psm.storeVPS(vps, nalu); }
        if (type == NAL_UNIT_SPS) {
          TComSPS* s = new TComSPS(); dec->parseSPS(s);
          s->setBitsForPOC(bitsPoc); s->setMaxCUWidth(1u << log2Ctb); s->setMaxCUHeight(1u << log2Ctb); s->setUseSAO(sao != 0); s->setSPSTemporalMVPEnabledFlag(tmvp != 0);
          s->setLongTermRefsPresent(false); s->setMaxDecPicBuffering(6, 0);
          s->createRPSList(nRps);
          for (int r = 0; r < nRps; r++) {
            TComReferencePictureSet* rps = s->getRPSList()->getReferencePictureSet(r);
            if (ctcSets) {
              const int np = r == 2 ? 2 : 1;
              rps->setInterRPSPrediction(false); rps->setNumberOfNegativePictures(np); rps->setNumberOfPositivePictures(0); rps->setNumberOfPictures(np);
              if (r == 2) { rps->setDeltaPOC(0, -1); rps->setDeltaPOC(1, -2); rps->setUsed(0, true); rps->setUsed(1, true); } else { rps->setDeltaPOC(0, -(r + 1)); rps->setUsed(0, true); }
              continue;
            }
            rps->setInterRPSPrediction(false); rps->setNumberOfNegativePictures(r + 1); rps->setNumberOfPositivePictures(0); rps->setNumberOfPictures(r + 1);
            for (int j = 0; j <= r; j++) { rps->setDeltaPOC(j, -(j + 1)); rps->setUsed(j, true); }
          }
          psm.storeSPS(s, nalu);
        }
        if (type == NAL_UNIT_PPS) { TComPPS* p = new TComPPS(); dec->parsePPS(p); psm.storePPS(p, nalu); }
        if (type < 32) {
          TComSlice slice; slice.initSlice(); slice.setNalUnitType((NalUnitType)type); slice.setTLayer(0);
          dec->parseSliceHeader(&slice, &psm, prevPoc);
          // a dependent slice segment carries nothing but its address: parseSliceHeader leaves the slice object as initSlice made it and the reference's
          // caller fills it from the previous segment (TDecTop, copySliceInfo); the same is done here with the values of the slice's independent segment
          static int v[16]; static char rpsText[128] = "[]"; static char wpText[512] = "[]";
          const int dep = slice.getDependentSliceSegmentFlag() ? 1 : 0;
          if (!dep) {
            v[0] = (int)slice.getSliceType(); v[1] = slice.getPOC(); v[2] = (int)slice.getEnableTMVPFlag(); v[3] = (int)slice.getSaoEnabledFlag(CHANNEL_TYPE_LUMA);
            v[4] = (int)slice.getSaoEnabledFlag(CHANNEL_TYPE_CHROMA); v[5] = slice.isIntra() ? 0 : slice.getNumRefIdx(REF_PIC_LIST_0); v[6] = (int)slice.getCabacInitFlag();
            v[7] = slice.isIntra() ? 0 : (int)slice.getColRefIdx(); v[8] = slice.isIntra() ? 0 : (int)slice.getMaxNumMergeCand(); v[9] = slice.getSliceQp();
            v[10] = slice.getSliceChromaQpDelta(COMPONENT_Cb); v[11] = slice.getSliceChromaQpDelta(COMPONENT_Cr); v[12] = (int)slice.getDeblockingFilterDisable();
            v[13] = slice.getDeblockingFilterBetaOffsetDiv2(); v[14] = slice.getDeblockingFilterTcOffsetDiv2(); v[15] = (int)slice.getLFCrossSliceBoundaryFlag();
            // the slice's short-term reference picture set as parseSliceHeader left it (from the SPS by index, or parseShortTermRefPicSet on the header's own set)
            int o = snprintf(rpsText, sizeof(rpsText), "[");
            const TComReferencePictureSet* rps = slice.getRPS();
            if (!slice.getIdrPicFlag() && rps) for (int j = 0; j < rps->getNumberOfPictures() && j < 4; j++) o += snprintf(rpsText + o, sizeof(rpsText) - o, "%s[%d,%d]", j ? "," : "", rps->getDeltaPOC(j), (int)rps->getUsed(j));
            snprintf(rpsText + o, sizeof(rpsText) - o, "]");
            // pred_weight_table as xParsePredWeightTable (:2094) left it: denominators, then per RefPicList0 entry the flags and weight / offset per component
            const TComPPS* pp = psm.getPPS(slice.getPPSId());
            if (pp && pp->getUseWP() && slice.getSliceType() == P_SLICE) {
              WPScalingParam* wp = NULL; slice.getWpScaling(REF_PIC_LIST_0, 0, wp);
              int q = snprintf(wpText, sizeof(wpText), "[%d,%d", (int)wp[0].uiLog2WeightDenom, (int)wp[1].uiLog2WeightDenom);
              for (int r = 0; r < slice.getNumRefIdx(REF_PIC_LIST_0) && r < 4; r++) {
                slice.getWpScaling(REF_PIC_LIST_0, r, wp);
                q += snprintf(wpText + q, sizeof(wpText) - q, ",[%d,%d,%d,%d,%d,%d,%d,%d]", (int)wp[0].bPresentFlag, (int)wp[1].bPresentFlag, wp[0].iWeight, wp[0].iOffset, wp[1].iWeight, wp[1].iOffset, wp[2].iWeight, wp[2].iOffset);
              }
              snprintf(wpText + q, sizeof(wpText) - q, "]");
            } else snprintf(wpText, sizeof(wpText), "[]");
          }
          printf("%s{\"nal_type\":%d,\"address\":%d,\"dependent\":%d,\"slice_type\":%d,\"poc\":%d,\"tmvp\":%d,\"sao_luma\":%d,\"sao_chroma\":%d,\"num_ref_idx\":%d,\"cabac_init\":%d,"
                 "\"col_ref_idx\":%d,\"max_merge_cand\":%d,\"qp\":%d,\"cb_qp_offset\":%d,\"cr_qp_offset\":%d,\"deblocking_disabled\":%d,\"beta_offset_div2\":%d,"
                 "\"tc_offset_div2\":%d,\"lf_across\":%d,\"rps\":%s,\"wp\":%s}\n", n++ ? "," : "", type, (int)slice.getSliceSegmentCurStartCtuTsAddr(), dep, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9],
                 v[10], v[11], v[12], v[13], v[14], v[15], rpsText, wpText);
          const bool keepsAnchor = (type <= 14 && (type & 1) == 0) || (type >= 6 && type <= 9);   // 8.3.1: RASL / RADL / sub-layer non-reference pictures are not prevTid0Pic
          if (dep) { if (!keepsAnchor) prevPoc = v[1]; if (i < size) { sc = data[i + 2] == 0 ? 4 : 3; index = i; i += sc; } continue; }
          if (!keepsAnchor) prevPoc = slice.getPOC();
        }
        if (i < size) { sc = data[i + 2] == 0 ? 4 : 3; index = i; i += sc; }
      }
    }
    printf("]\n");
    return 0;
  }
  fprintf(stderr, "usage: %s tables | hls <annexb> | slices <annexb> <log2_max_poc_lsb> <log2_ctb> <sao> <tmvp> <num_st_rps> [ctc_sets]\n", argv[0]);
  return 1;
}
