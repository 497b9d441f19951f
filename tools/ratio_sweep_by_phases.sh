# best split per phase count P: P = 2 (43 648 messages), 4 (52 416), 6 (55 296), and 38 912 (P = 2, more spare lanes)
for B in 43648 52416 55296 38912; do
B=$B RATIOS="1.44 1.47 1.50 1.53 1.56" BATCHES="" bash tools/ratio_sweep.sh 2>&1
done
