// ethcnn_io.cpp -- the small text/config pieces of the driver.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ethcnn_spec.h"

namespace ethcnn {

// net_CNN.get_thresholds (/root/reference/HM-16.5_Test_AI/bin/net_CNN.py:38-47):
//   line = f.readline(); str_arr = line.split(' ');
//   thr_l1_lower = float(str_arr[1]); thr_l2_lower = float(str_arr[3])
// i.e. split on SINGLE spaces (empty tokens count), float() strips surrounding whitespace.
int parse_thr_info(const char* path, float* thr1, float* thr2, char* err, size_t errcap) {
    FILE* f = std::fopen(path, "r");
    if (!f) {
        std::snprintf(err, errcap, "cannot open %s", path);
        return ETHCNN_ERR_IO;
    }
    std::string line;
    int ch;
    while ((ch = std::fgetc(f)) != EOF) {
        line.push_back((char)ch);
        if (ch == '\n') break;
    }
    std::fclose(f);
    std::vector<std::string> tok;
    size_t start = 0;
    for (;;) {
        const size_t sp = line.find(' ', start);
        if (sp == std::string::npos) {
            tok.push_back(line.substr(start));
            break;
        }
        tok.push_back(line.substr(start, sp - start));
        start = sp + 1;
    }
    auto to_float = [](const std::string& s, float* out) {
        size_t a = 0, b = s.size();
        while (a < b && std::isspace((unsigned char)s[a])) ++a;
        while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
        if (a == b) return false;
        const std::string t = s.substr(a, b - a);
        char* end = nullptr;
        const double v = std::strtod(t.c_str(), &end);
        if (end != t.c_str() + t.size()) return false;
        *out = (float)v;
        return true;
    };
    if (tok.size() < 4 || !to_float(tok[1], thr1) || !to_float(tok[3], thr2)) {
        std::snprintf(err, errcap, "%s: first line must hold >= 4 space-separated floats (tokens [1],[3] are used)", path);
        return ETHCNN_ERR_FORMAT;
    }
    return 0;
}

}  // namespace ethcnn

// video_to_cu_depth.py:126-133
extern "C" int ethcnn_model_name_for_qp(int qp, char* out, size_t cap) {
    const char* name = qp < 25 ? "model_2000000_qp20~25.dat"
                     : qp < 30 ? "model_2000000_qp25~30.dat"
                     : qp < 35 ? "model_2000000_qp30~35.dat"
                               : "model_2000000_qp35~40.dat";
    if (!out || cap <= std::strlen(name)) return ETHCNN_ERR_ARG;
    std::strcpy(out, name);
    return ETHCNN_OK;
}

// resi_to_cu_depth_LDP.py:170-177 (LSTM file per QP band); the CNN file is fixed (:160)
extern "C" int ethcnn_lstm_model_name_for_qp(int qp, char* out, size_t cap) {
    const char* name = qp < 25 ? "model_LDP_200000_qp22.dat"
                     : qp < 30 ? "model_LDP_200000_qp27.dat"
                     : qp < 35 ? "model_LDP_200000_qp32.dat"
                               : "model_LDP_200000_qp37.dat";
    if (!out || cap <= std::strlen(name)) return ETHCNN_ERR_ARG;
    std::strcpy(out, name);
    return ETHCNN_OK;
}

extern "C" int ethcnn_parse_thresholds(const char* path, float* thr_l1_lower, float* thr_l2_lower) {
    if (!path || !thr_l1_lower || !thr_l2_lower) return ETHCNN_ERR_ARG;
    char err[400];
    return ethcnn::parse_thr_info(path, thr_l1_lower, thr_l2_lower, err, sizeof err);
}
