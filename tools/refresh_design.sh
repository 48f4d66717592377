#!/bin/bash
# DESIGN.md = tools/design_front.md filled from profiles/<tag>_* + Appendix A of the current DESIGN.md:  tools/refresh_design.sh [tag]
TAG=${1:-r05b}
cd "$(dirname "$0")/.."
python3 tools/fill_design.py $TAG --targets "met for cfg3 fp8, cfg5 and cfg5 fp8; cfg4 sits ON the line (159.8 on the box of this record; 160.6 – 162.5 in the same-box A/Bs of the last kernel change on two other boxes — the day's boxes differ by ± 2 %); missed for cfg3 in bf16 (308; 312.6 on the day's fastest box)" > /tmp/front_filled.md && python3 tools/splice_design.py /tmp/front_filled.md
