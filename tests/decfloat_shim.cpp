// test shim: csrc/decfloat.hpp compiled for the host (tests/test_decfloat.py)
#include "../tao_amodal_amd/csrc/decfloat.hpp"
extern "C" int decf_parse(const char *s, int n, double *out, int *is_int, long long *iv)
{
    bool b = false;
    int64_t v = 0;
    const int used = decf::parse_json_number([s](int i) { return (unsigned char)s[i]; }, n, *out, &b, &v);
    *is_int = b;
    *iv = v;
    return used;
}
// many numbers at once: text = strings joined by '\n'; used[k] = bytes used (0 = undecided)
extern "C" void decf_parse_many(const char *text, const long long *off, long long n,
                                double *out, int *used)
{
    for (long long k = 0; k < n; k++) {
        const char *s = text + off[k];
        const int len = (int)(off[k + 1] - off[k] - 1);
        used[k] = decf::parse_json_number([s](int i) { return (unsigned char)s[i]; }, len, out[k]);
    }
}
