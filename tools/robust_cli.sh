# prints rc= per case instead of stopping at the first failure
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && rm -rf rb && mkdir rb && cd rb
G=$ROOT/tests/golden
B=$ROOT/soap3-dp_amd/soap3-dp
for h in 1 2 3 4; do $B pair $G/g4/index $G/g4/reads_1.fq $G/g4/reads_2.fq -u 500 -v 100 -L 100 -h $h -o p$h > p$h.log 2>&1; echo "pair -h $h rc=$? gout=$(grep -vc '^@' p$h.gout.1) dpout=$(grep -vc '^@' p$h.dpout.1) unpair=$(grep -vc '^@' p$h.unpair)"; done
gzip -c $G/g4/reads_1.fq > r1.fq.gz; gzip -c $G/g4/reads_2.fq > r2.fq.gz
$B pair $G/g4/index r1.fq.gz r2.fq.gz -u 500 -v 100 -L 100 -o pgz > pgz.log 2>&1; echo "gz rc=$?"; cmp <(grep -v '^@PG' pgz.gout.1 | sed 's/r1.fq.gz/X/g') <(grep -v '^@PG' p2.gout.1 | sed "s#$G/g4/reads_1.fq#X#g") && echo "gz == plain (gout)"
awk 'NR%4==1{print ">"substr($0,2)} NR%4==2{print}' $G/g4/reads_1.fq > r1.fa; awk 'NR%4==1{print ">"substr($0,2)} NR%4==2{print}' $G/g4/reads_2.fq > r2.fa
$B pair $G/g4/index r1.fa r2.fa -u 500 -v 100 -L 100 -o pfa > pfa.log 2>&1; echo "fasta rc=$? records=$(cat pfa.gout.1 pfa.dpout.1 pfa.unpair | grep -vc '^@')"
for h in 1 3 4; do $B single $G/g2/index $G/g2/reads.fq -L 120 -h $h -o s$h > s$h.log 2>&1; echo "single default -h $h rc=$? gout=$(grep -vc '^@' s$h.gout.1) dpout=$(grep -vc '^@' s$h.dpout.1)"; done
$B single $G/g3/index $G/g3/reads.fq -L 200 -o s200 > s200.log 2>&1; echo "single -L 200 default rc=$? gout=$(grep -vc '^@' s200.gout.1) dpout=$(grep -vc '^@' s200.dpout.1)"
