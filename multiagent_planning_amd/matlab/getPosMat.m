function Apos = getPosMat(h,K)
% Shadows dmpc/matlab/getPosMat.m (same signature): Lambda (3K x 3K), acceleration inputs -> positions, from the
% library's host routine dmpc_model_matrices (bit-identical to the reference's recurrence).
prm = dmpc_params_struct(0, h, K, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);
Lt = dmpc_mex('model_matrices', prm);      % row-major C matrix in a column-major array: the transpose
Apos = Lt';
end
