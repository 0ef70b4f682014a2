#!/bin/bash
# Keep the CURRENT build as the "old" side of tools/ab_build.sh (tools/_ab_old/, git-ignored):  bash tools/ab_snapshot.sh
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf $R/tools/_ab_old && mkdir -p $R/tools/_ab_old/profiles
cp -r $R/bench.py $R/models.py $R/include $R/oracle $R/BASELINE.json $R/tests $R/tools/_ab_old/
cp $R/profiles/pmc_moment.json $R/tools/_ab_old/profiles/
mkdir -p $R/tools/_ab_old/video-moment-localization_amd
cp $R/video-moment-localization_amd/*.py $R/video-moment-localization_amd/*.so $R/tools/_ab_old/video-moment-localization_amd/
mkdir -p $R/tools/_ab_old/video-moment-localization_amd/csrc
cp $R/video-moment-localization_amd/csrc/gemm.h $R/video-moment-localization_amd/csrc/moment_unit.hip $R/tools/_ab_old/video-moment-localization_amd/csrc/
du -sh $R/tools/_ab_old
