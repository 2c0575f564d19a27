function Delta = getDeltaMat(k_hor)
% Shadows dmpc/matlab/getDeltaMat.m (same signature): first block I, then [-I I] bidiagonal.
prm = dmpc_params_struct(0, 0.2, k_hor, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);
[~,~,~,Dt] = dmpc_mex('model_matrices', prm);
Delta = Dt';
end
