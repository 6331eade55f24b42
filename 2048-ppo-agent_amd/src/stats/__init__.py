from .running_stats_vec import RunningStatsVec
