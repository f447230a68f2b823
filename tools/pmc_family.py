"""Print (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch for the conv / GEMM families from a tools/pmc_family.sh output directory."""
import json
import sys

out, tag = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
fe, wr = json.load(open(out + "/pmc_fetch.json")), json.load(open(out + "/pmc_write.json"))
for k in ("k_conv_igemm<128,128>", "k_conv_igemm<64,64>", "k_conv_halo", "k_conv_splitk_reduce", "k_gemm_pw"):
    if k in fe and k in wr:
        f, w = 2 * fe[k].get("FETCH_SIZE", 0.0) * 1024 / 1e6, wr[k].get("WRITE_SIZE", 0.0) * 1024 / 1e6
        print("%s %-24s %6.1f MB/launch (fetch %.1f + write %.1f)  %3d launches  %.1f us" % (tag, k, f + w, f, w, fe[k]["launches"], fe[k]["avg_us"]))
