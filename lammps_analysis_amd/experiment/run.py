"""Dispatch of calculators and transformations.

``experiment.run.<Name>`` / ``project.run.<Name>`` return a fresh calculator bound
to the experiment(s), or a wrapper that runs a transformation on them, as
mdsuite/experiment/run.py:58-243 of the reference does.  Only the calculators and
transformations of the accelerated path exist in this engine.
"""
from __future__ import annotations

import functools
from typing import List


class RunComputation:
    def __init__(self, experiment=None, experiments: List = None):
        self.experiment = experiment
        self.experiments = experiments
        self.kwargs = {"experiment": experiment, "experiments": experiments}

    def exp_wrapper(self, func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            return func(*args, **kwargs)

        return wrapper

    def transformation_wrapper(self, func):
        """Run the transformation on every selected experiment."""

        @functools.wraps(func.run_transformation)
        def wrapper(*args, **kwargs):
            if self.experiments is None:
                self.experiments = [self.experiment]
            for experiment in self.experiments:
                experiment.cls_transformation_run(func(), *args, **kwargs)

        return wrapper

    # ------------------------------------------------------- transformations

    @property
    def IonicCurrent(self):
        from ..transformations import IonicCurrent
        return self.transformation_wrapper(IonicCurrent)

    @property
    def TranslationalDipoleMoment(self):
        from ..transformations import TranslationalDipoleMoment
        return self.transformation_wrapper(TranslationalDipoleMoment)

    @property
    def MomentumFlux(self):
        from ..transformations import MomentumFlux
        return self.transformation_wrapper(MomentumFlux)

    @property
    def IntegratedHeatCurrent(self):
        from ..transformations import IntegratedHeatCurrent
        return self.transformation_wrapper(IntegratedHeatCurrent)

    @property
    def KinaciIntegratedHeatCurrent(self):
        from ..transformations import KinaciIntegratedHeatCurrent
        return self.transformation_wrapper(KinaciIntegratedHeatCurrent)

    @property
    def ThermalFlux(self):
        from ..transformations import ThermalFlux
        return self.transformation_wrapper(ThermalFlux)

    @property
    def CoordinateWrapper(self):
        from ..transformations import CoordinateWrapper
        return self.transformation_wrapper(CoordinateWrapper)

    @property
    def CoordinateUnwrapper(self):
        from ..transformations import CoordinateUnwrapper
        return self.transformation_wrapper(CoordinateUnwrapper)

    @property
    def ScaleCoordinates(self):
        from ..transformations import ScaleCoordinates
        return self.transformation_wrapper(ScaleCoordinates)

    @property
    def UnwrapViaIndices(self):
        from ..transformations import UnwrapViaIndices
        return self.transformation_wrapper(UnwrapViaIndices)

    @property
    def VelocityFromPositions(self):
        from ..transformations import VelocityFromPositions
        return self.transformation_wrapper(VelocityFromPositions)

    # ----------------------------------------------------------- calculators

    def _calculator(self, name: str):
        from .. import calculators
        return self.exp_wrapper(getattr(calculators, name))(**self.kwargs)

    @property
    def CoordinationNumbers(self):
        return self._calculator("CoordinationNumbers")

    @property
    def PotentialOfMeanForce(self):
        return self._calculator("PotentialOfMeanForce")

    @property
    def KirkwoodBuffIntegral(self):
        return self._calculator("KirkwoodBuffIntegral")

    @property
    def AngularDistributionFunction(self):
        return self._calculator("AngularDistributionFunction")

    @property
    def RadialDistributionFunction(self):
        return self._calculator("RadialDistributionFunction")

    @property
    def GreenKuboIonicConductivity(self):
        return self._calculator("GreenKuboIonicConductivity")

    @property
    def GreenKuboDiffusionCoefficients(self):
        return self._calculator("GreenKuboDiffusionCoefficients")

    @property
    def GreenKuboViscosity(self):
        return self._calculator("GreenKuboViscosity")

    @property
    def GreenKuboThermalConductivity(self):
        return self._calculator("GreenKuboThermalConductivity")

    @property
    def GreenKuboViscosityFlux(self):
        return self._calculator("GreenKuboViscosityFlux")

    @property
    def EinsteinDiffusionCoefficients(self):
        return self._calculator("EinsteinDiffusionCoefficients")

    @property
    def EinsteinDistinctDiffusionCoefficients(self):
        return self._calculator("EinsteinDistinctDiffusionCoefficients")

    @property
    def GreenKuboDistinctDiffusionCoefficients(self):
        return self._calculator("GreenKuboDistinctDiffusionCoefficients")

    @property
    def EinsteinHelfandThermalConductivity(self):
        return self._calculator("EinsteinHelfandThermalConductivity")

    @property
    def NernstEinsteinIonicConductivity(self):
        return self._calculator("NernstEinsteinIonicConductivity")

    @property
    def EinsteinHelfandThermalKinaci(self):
        return self._calculator("EinsteinHelfandThermalKinaci")

    @property
    def EinsteinHelfandIonicConductivity(self):
        return self._calculator("EinsteinHelfandIonicConductivity")
