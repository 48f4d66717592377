#!/bin/bash
# DESIGN.md = tools/design_front.md filled from profiles/<tag>_* + Appendix A of the current DESIGN.md:  tools/refresh_design.sh [tag]
TAG=${1:-r05b}
cd "$(dirname "$0")/.."
python3 tools/fill_design.py $TAG > /tmp/front_filled.md && python3 tools/splice_design.py /tmp/front_filled.md
