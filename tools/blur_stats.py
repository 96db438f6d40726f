import csv, sys
tot = {}
for r in csv.reader(open(sys.argv[1])):
    if 'gauss_blur' in r[0]:
        name = r[0].split('(')[0].replace('void mods::', '').replace('mods::', '')
        print('%-34s calls %4s avg %7.2f us' % (name, r[1], float(r[3]) / 1000))
        tot[0] = tot.get(0, 0) + float(r[2]); tot[1] = tot.get(1, 0) + int(r[1])
    if 'hessian_response' in r[0] or 'nms_kernel' in r[0] or 'resize_half' in r[0]:
        print('%-34s calls %4s avg %7.2f us total %8.1f us' % (r[0].split('(')[0].replace('mods::', ''), r[1], float(r[3]) / 1000, float(r[2]) / 1000))
print('blur total %.1f us over %d launches' % (tot[0] / 1000, tot[1]))
