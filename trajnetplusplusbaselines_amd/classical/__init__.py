from . import constant_velocity, kalman, orca, socialforce
