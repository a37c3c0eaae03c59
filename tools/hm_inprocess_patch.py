#!/usr/bin/env python3
"""Applies the in-process predictor hook to a COPY of the reference encoder source.

    tools/hm_inprocess_patch.py <copy of HM-16.5_Test_AI/source>

Two edits, both at the reference's own hook sites:

  App/TAppEncoder/TAppEncCfg.cpp:2317-2321   the `system("python video_to_cu_depth.py ...")` call is REMOVED
      (nothing is predicted before encoding, the YUV file is not read a second time);
  Lib/TLibEncoder/TEncCu.cpp:237-261         at the first CTU of every picture, instead of `fread`ing nCtu*21
      floats from cu_depth.dat, compressCtu passes the picture's own luma plane (TComPicYuv* getPicYuvOrg():
      Pel samples, stride) to `ethcnn_hm_predict_picture` (tools/hm_inprocess_hook.c -> libethcnn.so) and
      stores the result with TComPic::setCUDepth exactly as before (TComPic.h:84-88).

Because only the pictures HM encodes are predicted, FrameSkip / FramesToBeEncoded behave correctly (the
file-based reference predicts all frames from 0 and reads cu_depth.dat from its start regardless,
video_to_cu_depth.py:139-140).  The patch is kept out of the reference tree: oracle/build_ref_hm.sh applies it
to its temporary copy only.
"""
import re
import sys


def patch_cfg(path):
    s = open(path, encoding="latin1").read()
    pat = re.compile(r'[ \t]*sprintf\(cmd, "python video_to_cu_depth\.py[^\n]*\n[ \t]*printf\("%s\\n", cmd\);\n[ \t]*assert\(system\(cmd\)==0\);\n')
    assert len(pat.findall(s)) == 1, "TAppEncCfg.cpp: predictor call site not found"
    new = ('\tprintf("ethcnn: in-process predictor (per picture, from the encoder\'s luma buffers; no cu_depth.dat)\\n");\n'
           '\t(void)cmd;\n')
    s = pat.sub(lambda m: new, s)  # a callable: no escape processing of the replacement text
    open(path, "w", encoding="latin1").write(s)


def patch_cu(path):
    s = open(path, encoding="latin1").read()
    a = '  static FILE * fpCUDepth = fopen("cu_depth.dat", "rb");\n'
    assert s.count(a) == 1, "TEncCu.cpp: cu_depth.dat open not found"
    s = s.replace(a, "")
    pat = re.compile(r'[ \t]*assert\(fread\(pCUDepthTemp, sizeof\(float\), validWidthInCTU\*validHeightInCTU \* 21, fpCUDepth\)>0\);\n')
    assert len(pat.findall(s)) == 1, "TEncCu.cpp: per-picture fread not found"
    s = pat.sub(lambda m: (
        '\t  {\n'
        '\t    TComPicYuv* pcOrg_ = pCtu->getPic()->getPicYuvOrg();\n'
        '\t    static_assert(sizeof(Pel) == sizeof(short), "ethcnn_hm_predict_picture takes 16-bit samples");\n'
        '\t    if (ethcnn_hm_predict_picture(pcOrg_->getAddr(COMPONENT_Y), pcOrg_->getStride(COMPONENT_Y), iWidth, iHeight,\n'
        '\t                                  pCtu->getSlice()->getSPS()->getBitDepth(CHANNEL_TYPE_LUMA), m_pcEncCfg->getQP(), pCUDepthTemp) != 0)\n'
        '\t    {\n'
        '\t      fprintf(stderr, "ethcnn: in-process prediction failed\\n");\n'
        '\t      exit(1);\n'
        '\t    }\n'
        '\t  }\n'), s)
    decl = 'extern "C" int ethcnn_hm_predict_picture(const short* luma, int stride, int width, int height, int bit_depth, int qp, float* probs);\n'
    anchor = "Void TEncCu::compressCtu( TComDataCU* pCtu )"
    assert s.count(anchor) == 1
    i = s.rfind("/**", 0, s.index(anchor))  # put the declaration before the function's doc comment
    s = s[:i] + decl + s[i:]
    open(path, "w", encoding="latin1").write(s)


if __name__ == "__main__":
    src = sys.argv[1]
    patch_cfg(src + "/App/TAppEncoder/TAppEncCfg.cpp")
    patch_cu(src + "/Lib/TLibEncoder/TEncCu.cpp")
    print("in-process hook applied to", src)
