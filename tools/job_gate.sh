for nb in 83000 125000 170000 250000 400000; do
for o in 0 1; do python3 tools/shape_ab.py 200000 $nb 200 50 short_ord=$o dual_short_min_tiles=1 | cut -c1-120; done
done
for nb in 60000 125000 250000; do
for o in 0 1; do python3 tools/shape_ab.py 200000 $nb 200 26 short_ord=$o dual_short_min_tiles=1 | cut -c1-120; done
done
for nb in 125000 250000 400000; do
for o in 0 1; do python3 tools/shape_ab.py 100000 $nb 200 100 short_ord=$o dual_short_min_tiles=1 | cut -c1-120; done
done
