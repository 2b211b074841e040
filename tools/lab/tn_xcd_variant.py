"""Rewrites the block -> (group, row slice) mapping of k_gemm_tn16 in a COPY of gemm_bf16.hip so that the k groups of a row
slice land on one XCD (blocks b and b + 8):  python3 tools/lab/tn_xcd_variant.py /tmp/nt16lab/gemm_bf16_tnx.hip, then compile
tools/lab/nt16_lab.hip with -DKSRC pointing at it (see build_variant.sh) into tools/bin/nt16_tnx."""
import sys
src=open(sys.argv[1]).read()
old="    const int grp = blockIdx.x % per, set = blockIdx.x / per;\n"
new='''    int grp, set;
    {
        const int b = blockIdx.x, chunk = 8 * per, full = (a.sets / 8) * chunk;
        if (b < full) { grp = (b / 8) % per; set = (b / chunk) * 8 + (b & 7); }
        else { grp = (b - full) % per; set = (a.sets / 8) * 8 + (b - full) / per; }
    }
'''
assert old in src
open(sys.argv[1],'w').write(src.replace(old,new))
