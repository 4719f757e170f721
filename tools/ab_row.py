"""one line of an A/B table from a bench_detail.json: step, serial step, and the per-step time of the kernel families named in AB_KERNELS"""
import json
import os
import sys
d = json.load(open(sys.argv[1]))
ks = {k["name"]: k for k in d.get("single_stream_kernels", [])}
want = os.environ.get("AB_KERNELS", "convnext32_bwd_kernel<2,true>,convnext32_pass2_kernel<true>,convnext32_pass1_kernel<true>,"
                                    "conv32p_kernel<true>,wgradb16_kernel<3,true>,wgrad_reduce_multi_kernel,convp16_kernel<2,true>").split(",")
cols = "  ".join(f"{n.split('_kernel')[0]}{n.split('_kernel')[1] if '_kernel' in n else ''} {ks[n]['ms_per_step']:.2f}" for n in want if n in ks)
print(f"{sys.argv[2]:12s} step {d['ms_per_step']:.2f} ms  serial {d.get('single_stream_step_ms', 0):.2f} ms | {cols}   [{sys.argv[3] if len(sys.argv) > 3 else ''}]")
