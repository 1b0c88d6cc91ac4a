"""rows of a rocprofv3 kernel_stats.csv that belong to the optimisation step's own kernels (prof_train.sh)"""
import csv
import sys

for r in list(csv.reader(open(sys.argv[1])))[1:]:
    if any(k in r[0] for k in ("opt_", "pack_conv3", "param_cast", "seg_loss", "fill_")):
        print("%-62s calls %5s avg %7.1f us" % (r[0][:62], r[1], float(r[3]) / 1000))
