"""Runtime contract checks; same decorators as the reference's scripts/utils/validators.py
(the asserts ARE its error convention).  Only the ones the hot path's callers use."""


def validate_forward_pass_inputs(forward_pass_func):
    def func_wrapper(self, input_images, target_images, split, iteration, t_interp):
        n = self.cfg.getint("TRAIN", "N_FRAMES")
        assert input_images.shape[1] == n
        assert target_images.shape[1] == t_interp.shape[1] == n - 1
        assert bool((t_interp > 0).all() and (t_interp < 1).all()), "Interpolation values out of bounds."
        return forward_pass_func(self, input_images, target_images, split, iteration, t_interp)

    return func_wrapper


def validate_t_interp(t_interp_generator_func):
    def func_wrapper(self, mid_idx):
        t_interp = t_interp_generator_func(self, mid_idx)
        assert (t_interp > 0).all() and (t_interp < 1).all(), "Incorrect values."
        return t_interp

    return func_wrapper


def validate_evaluation_interpolation_result(interpolation_func):
    def func_wrapper(self, current_batch):
        outputs = interpolation_func(self, current_batch)
        if not self.dataset == "VIMEO":
            assert len(outputs) == self.interp_factor - 1, "Wrong number of outputs."
        return outputs

    return func_wrapper
