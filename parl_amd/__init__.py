"""parl_amd — the MI355X-native IMPALA / A2C actor-learner hot path behind PARL's API."""
__version__ = '0.1.0'
